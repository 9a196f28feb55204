// Micro-benchmark: which instruction streams of two waves on ONE SIMD overlap on gfx950?
//   f32 MFMA (v_mfma_f32_16x16x4_f32)  |  bf16 MFMA (v_mfma_f32_16x16x32_bf16)  |  v_pk_fma_f32  |  v_fma_f32
// 512-thread workgroups (waves w and w + 4 share a SIMD), one workgroup per CU (LDS), 256 registers per wave.
// mode bits: low nibble = what waves 0-3 run, high nibble = what waves 4-7 run (0 nothing, 1 f32 MFMA, 2 bf16 MFMA,
// 3 v_pk_fma_f32, 4 v_fma_f32).  Every stream is sized to the same single-wave duration (32 cycles per unit).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int WHAT>
__device__ __forceinline__ float run(int iters, float seed) {
    float out = 0.f;
    if constexpr (WHAT == 1) {
        f32x4 a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < iters; ++i) {              // 4 independent MFMAs = 128 cycles
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0f, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0f, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0f, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0f, a3, 0, 0, 0);
        }
        out = a0[0] + a1[1] + a2[2] + a3[3];
    } else if constexpr (WHAT == 2) {
        f32x4 a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        bf16x8 x, y;
        for (int k = 0; k < 8; ++k) { x[k] = (__bf16)seed; y[k] = (__bf16)1.0f; }
        for (int i = 0; i < iters; ++i) {              // 8 MFMAs of 16 cycles = 128 cycles
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
        }
        out = a0[0] + a1[1] + a2[2] + a3[3];
    } else if constexpr (WHAT == 3) {
        f32x2 v[8];
        for (int k = 0; k < 8; ++k) v[k] = f32x2{seed + k, seed - k};
        const f32x2 m = {1.0000001f, 0.9999999f}, c = {seed, -seed};
        for (int i = 0; i < iters; ++i) {              // 32 independent v_pk_fma_f32 (4 cycles each alone) = 128 cycles
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_elementwise_fma(v[k], m, c);
        }
        for (int k = 0; k < 8; ++k) out += v[k].x + v[k].y;
    } else if constexpr (WHAT == 4) {
        float v[8];
        for (int k = 0; k < 8; ++k) v[k] = seed + k;
        for (int i = 0; i < iters; ++i) {              // 32 independent v_fma_f32
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], 1.0000001f, seed);
        }
        for (int k = 0; k < 8; ++k) out += v[k];
    }
    return out;
}

template <int LO, int HI>
__global__ __launch_bounds__(512) void pipes_kernel(int iters, float seed, float* sink, unsigned long long* cycles) {
    extern __shared__ float lds[];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");                       // 256 registers: two waves per SIMD, one WG per CU
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r = wave < 4 ? run<LO>(iters, seed) : run<HI>(iters, seed);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (r == 12345.678f) sink[threadIdx.x] = r + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) atomicMax(&cycles[wave < 4 ? 0 : 1], t1 - t0);
}

template <int LO, int HI>
static void go(const char* name, int iters) {
    float* sink; unsigned long long* cyc;
    hipMalloc(&sink, 4096); hipMalloc(&cyc, 16); hipMemset(cyc, 0, 16);
    hipFuncSetAttribute((const void*)pipes_kernel<LO, HI>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    pipes_kernel<LO, HI><<<256, 512, 100 * 1024>>>(iters, 1.0f, sink, cyc);      // warm
    hipMemset(cyc, 0, 16);
    hipEventRecord(e0);
    pipes_kernel<LO, HI><<<256, 512, 100 * 1024>>>(iters, 1.0f, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-44s %8.3f ms   cycles/unit: waves0-3 %6.1f  waves4-7 %6.1f\n", name, ms, (double)h[0] / iters / 4, (double)h[1] / iters / 4);
    hipFree(sink); hipFree(cyc);
}

int main() {
    const int it = 20000;     // units of 128 single-wave cycles
    go<1, 0>("f32 MFMA alone", it);
    go<2, 0>("bf16 MFMA alone", it);
    go<3, 0>("v_pk_fma_f32 alone", it);
    go<4, 0>("v_fma_f32 alone", it);
    go<1, 1>("f32 MFMA | f32 MFMA", it);
    go<3, 3>("v_pk_fma_f32 | v_pk_fma_f32", it);
    go<4, 4>("v_fma_f32 | v_fma_f32", it);
    go<1, 3>("f32 MFMA | v_pk_fma_f32", it);
    go<1, 4>("f32 MFMA | v_fma_f32", it);
    go<2, 3>("bf16 MFMA | v_pk_fma_f32", it);
    go<2, 4>("bf16 MFMA | v_fma_f32", it);
    go<2, 1>("bf16 MFMA | f32 MFMA", it);
    return 0;
}
