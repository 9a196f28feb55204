// Micro-benchmark 3: issue rate of v_mfma_f32_16x16x32_bf16 (and v_mfma_f32_16x16x4_f32) when consecutive instructions
// accumulate into the SAME register (dependency through SrcC) against 2 / 4 / 6 independent accumulators in rotation.
// One wave per SIMD (256 threads), 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int WAYS, bool BF16>
__global__ __launch_bounds__(256) void dep_kernel(int iters, float seed, float* sink, unsigned long long* cycles) {
    f32x4 acc[WAYS];
    for (int w = 0; w < WAYS; ++w) acc[w] = f32x4{seed, 0, 0, 0};
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)seed; b[k] = (__bf16)1.0f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 12 / WAYS; ++r)
#pragma unroll
            for (int w = 0; w < WAYS; ++w) {
                if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[w]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[w]) : "v"(seed), "v"(1.0f));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int w = 0; w < WAYS; ++w) r += acc[w][0];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);
}

template <int WAYS, bool BF16>
static void go(int it) {
    float* sink; unsigned long long* cyc;
    (void)hipMalloc(&sink, 4096); (void)hipMalloc(&cyc, 8); (void)hipMemset(cyc, 0, 8);
    dep_kernel<WAYS, BF16><<<256, 256>>>(it, 1.0f, sink, cyc);
    (void)hipMemset(cyc, 0, 8);
    dep_kernel<WAYS, BF16><<<256, 256>>>(it, 1.0f, sink, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s, %d accumulator(s) in rotation: %5.1f cycles per instruction\n", BF16 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x4_f32  ", WAYS,
           (double)h / it / 12);
    (void)hipFree(sink); (void)hipFree(cyc);
}

int main() {
    const int it = 20000;
    go<1, true>(it); go<2, true>(it); go<3, true>(it); go<4, true>(it); go<6, true>(it);
    go<1, false>(it); go<2, false>(it); go<4, false>(it);
    return 0;
}
