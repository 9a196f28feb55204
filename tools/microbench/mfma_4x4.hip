// Micro-benchmark 4 (round 6): v_mfma_f32_4x4x1_16b_f32 - sixteen independent 4x4 rank-1 updates per instruction, the f32
// matrix shape that fits the F = 19, d = 8 attention of the fused backward (layer_fused.hip phases D / E) without padding:
//   (1) fragment layout, checked against a scalar model: A lane 4b + i = A_b[i], B lane 4b + j = B_b[j],
//       D register i of lane 4b + j = D_b[i][j];
//   (2) issue rate with 1 / 2 / 5 accumulators in rotation (dependent and independent SrcC);
//   (3) the same with one ds_read_b64 (operand of the NEXT instruction) per instruction, operands through a register ring.
// One wave per SIMD (256 threads), 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

__global__ void layout_kernel(const float* a, const float* b, float* d) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[64 + threadIdx.x], b[64 + threadIdx.x], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[4 * threadIdx.x + r] = acc[r];
}

template <int WAYS>
__global__ __launch_bounds__(256) void rate_kernel(int iters, float seed, float* sink, unsigned long long* cycles) {
    f32x4 acc[WAYS];
    for (int w = 0; w < WAYS; ++w) acc[w] = f32x4{seed, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 20 / WAYS; ++r)
#pragma unroll
            for (int w = 0; w < WAYS; ++w)
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[w]) : "v"(seed), "v"(1.0f));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int w = 0; w < WAYS; ++w) r += acc[w][0];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);
}

// 20 steps per iteration: step s multiplies the pair read for it one step earlier (A = .x, both accumulators' B = own value)
template <int WAYS>
__global__ __launch_bounds__(256) void lds_kernel(int iters, float seed, float* sink, unsigned long long* cycles) {
    __shared__ float buf[64 * 36];
    for (int i = threadIdx.x; i < 64 * 36; i += 256) buf[i] = seed * (float)(i & 7);
    __syncthreads();
    f32x4 acc[WAYS];
    for (int w = 0; w < WAYS; ++w) acc[w] = f32x4{seed, 0, 0, 0};
    const float* base = buf + ((threadIdx.x >> 2) & 3) * 8 + 2 * (threadIdx.x & 3);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        f32x2 v[20];
#pragma unroll
        for (int s = 0; s < 20; ++s) v[s] = *reinterpret_cast<const f32x2*>(base + s * 36);
#pragma unroll
        for (int s = 0; s < 20; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(v[s].x, seed, acc[0], 0, 0, 0);
            acc[1 % WAYS] = __builtin_amdgcn_mfma_f32_4x4x1f32(v[s].y, seed, acc[1 % WAYS], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int w = 0; w < WAYS; ++w) r += acc[w][0];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);
}

template <typename K>
static double timed(K kern, int it) {
    float* sink; unsigned long long* cyc;
    (void)hipMalloc(&sink, 4096); (void)hipMalloc(&cyc, 8); (void)hipMemset(cyc, 0, 8);
    kern<<<256, 256>>>(it, 1.0f, sink, cyc);
    (void)hipMemset(cyc, 0, 8);
    kern<<<256, 256>>>(it, 1.0f, sink, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    (void)hipFree(sink); (void)hipFree(cyc);
    return (double)h / it;
}

int main() {
    // ---- layout -------------------------------------------------------------------------------------------------
    float ha[128], hb[128], hd[256], *da, *db, *dd;
    srand(1);
    for (int i = 0; i < 128; ++i) { ha[i] = (float)(rand() % 17 - 8); hb[i] = (float)(rand() % 13 - 6); }
    (void)hipMalloc(&da, sizeof ha); (void)hipMalloc(&db, sizeof hb); (void)hipMalloc(&dd, sizeof hd);
    (void)hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    layout_kernel<<<1, 64>>>(da, db, dd);
    (void)hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int blk = 0; blk < 16; ++blk)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const float want = ha[4 * blk + i] * hb[4 * blk + j] + ha[64 + 4 * blk + i] * hb[64 + 4 * blk + j];
                if (hd[4 * (4 * blk + j) + i] != want) ++bad;
            }
    printf("layout A lane 4b+i, B lane 4b+j, D reg i of lane 4b+j: %s (%d mismatches of 256)\n", bad ? "WRONG" : "confirmed", bad);
    // ---- rates --------------------------------------------------------------------------------------------------
    const int it = 20000;
    printf("v_mfma_f32_4x4x1_16b_f32, 1 accumulator : %5.1f cycles per instruction\n", timed(rate_kernel<1>, it) / 20);
    printf("v_mfma_f32_4x4x1_16b_f32, 2 accumulators: %5.1f cycles per instruction\n", timed(rate_kernel<2>, it) / 20);
    printf("v_mfma_f32_4x4x1_16b_f32, 4 accumulators: %5.1f cycles per instruction\n", timed(rate_kernel<4>, it) / 20);
    printf("v_mfma_f32_4x4x1_16b_f32, 5 accumulators: %5.1f cycles per instruction\n", timed(rate_kernel<5>, it) / 20);
    printf("20 ds_read_b64 + 40 MFMAs on 1 accumulator : %5.1f cycles per MFMA\n", timed(lds_kernel<1>, it) / 40);
    printf("20 ds_read_b64 + 40 MFMAs on 2 accumulators: %5.1f cycles per MFMA\n", timed(lds_kernel<2>, it) / 40);
    return 0;
}
