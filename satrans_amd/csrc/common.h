// Shared host/device helpers for libsatrans_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/satrans_hip.h"

namespace satrans {

constexpr int kWave = 64;

// Workgroup-range groups of the fixed-order reductions (fused_reduce_kernel / fused_reduce_all_kernel / head_reduce_kernel): a block
// is 32 elements x kReduceGroups groups, every group adds its contiguous share of the partial rows in index order, the group sums are
// combined in group order.  ONE constant for all of them: the head's rows are reduced by head_reduce_kernel or inside
// fused_reduce_all_kernel and must come out the same bits.  (A/B on one box, round 4: 4 / 8 / 16 / 32 groups -> 49 / 30 / 25 / 33 us
// for the three-layer reduction.)
constexpr int kReduceGroups = 16;

// thread-local description of the last failure, surfaced by satrans_last_error()
void set_error(const char* fmt, ...);

#define SATRANS_REQUIRE(cond, code, ...)      \
    do {                                      \
        if (!(cond)) {                        \
            ::satrans::set_error(__VA_ARGS__); \
            return (code);                    \
        }                                     \
    } while (0)

#define SATRANS_CHECK_LAUNCH(what)                                                          \
    do {                                                                                    \
        hipError_t e__ = hipGetLastError();                                                 \
        if (e__ != hipSuccess) {                                                            \
            ::satrans::set_error("%s: %s", (what), hipGetErrorString(e__));                 \
            return SATRANS_E_LAUNCH;                                                        \
        }                                                                                   \
    } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Read element `col` of sample `b` of X as an integer id (reference: `.long()` truncation of the fp32 id).
__device__ __forceinline__ int64_t load_id(const void* X, int id_dtype, int64_t x_stride, int64_t b, int col) {
    const int64_t at = b * x_stride + col;
    if (id_dtype == SATRANS_ID_F32) return (int64_t)((const float*)X)[at];
    if (id_dtype == SATRANS_ID_I32) return (int64_t)((const int32_t*)X)[at];
    return ((const int64_t*)X)[at];
}

// Row (sample b, field f) of a layer's input: the [B,F,D] activation, or - gather fused into the first layer - the arena
// row of that id.
__device__ __forceinline__ const float* layer_x_row(const satrans_layer_desc& a, int b, int f, int F, int D) {
    const size_t at = (size_t)b * F + f;
    return a.x_rows ? a.x + (size_t)a.x_rows[at] * D : a.x + at * D;
}

// Dense (float) column of X.  Integer id matrices carry no dense columns.
__device__ __forceinline__ float load_dense(const void* X, int64_t x_stride, int64_t b, int col) {
    return ((const float*)X)[b * x_stride + col];
}

}  // namespace satrans
