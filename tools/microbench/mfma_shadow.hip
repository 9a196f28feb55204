// Micro-benchmark 2: does anything issue in the shadow of a v_mfma_f32_16x16x4_f32 on gfx950?
//   (a) SAME wave: one MFMA followed by N independent VALU instructions of a kind, N = 0..8 - is the time per group
//       max(32, N * c) (shadow) or 32 + N * c (serial)?
//   (b) OTHER wave of the SIMD: an MFMA-only wave next to a wave of integer / transcendental / LDS instructions.
// One wave per SIMD for (a) (256 threads), two for (b) (512 threads, waves w and w + 4 share a SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// KIND: 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_xor_b32 (integer), 3 v_mul_lo_u32, 4 v_exp_f32, 5 v_cndmask_b32, 6 ds_read_b128, 7 v_mov_b32 dpp
template <int KIND>
__device__ __forceinline__ void valu(f32x2& p, float& s, uint32_t& u, f32x4& l, const float* lp) {
    if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(f32x2{1.0000001f, 0.9999999f}));
    else if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s) : "v"(1.0000001f));
    else if constexpr (KIND == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u) : "v"(0x9E3779B9u));
    else if constexpr (KIND == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u) : "v"(0x9E3779B1u));
    else if constexpr (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(s));
    else if constexpr (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(0x12345u));
    else if constexpr (KIND == 6) asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"((uint32_t)(uintptr_t)lp));
    else if constexpr (KIND == 7) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s));
}

template <int KIND, int N, bool MFMA>
__global__ __launch_bounds__(256) void shadow_kernel(int iters, float seed, float* sink, unsigned long long* cycles) {
    extern __shared__ float lds[];
    f32x4 a[4] = {{seed, 0, 0, 0}, {seed, 0, 0, 0}, {seed, 0, 0, 0}, {seed, 0, 0, 0}};
    f32x2 p[8]; float s[8]; uint32_t u[8]; f32x4 l[8];
    for (int k = 0; k < 8; ++k) { p[k] = f32x2{seed + k, seed - k}; s[k] = seed * 1e-3f + k; u[k] = threadIdx.x + k; l[k] = f32x4{0, 0, 0, 0}; }
    const float* lp = lds + 4 * threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if constexpr (MFMA) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a[m]) : "v"(seed), "v"(1.0f));
#pragma unroll
            for (int k = 0; k < N; ++k) valu<KIND>(p[k], s[k], u[k], l[k], lp);
        }
        if constexpr (KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = a[0][0] + a[1][1] + a[2][2] + a[3][3];
    for (int k = 0; k < 8; ++k) r += p[k].x + p[k].y + s[k] + (float)u[k] + l[k][0];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);
}

// (b) waves 0-3: MFMA only; waves 4-7: VALU kind only (N = 8 per group)
template <int KIND, bool MFMA>
__global__ __launch_bounds__(512) void cross_kernel(int iters, int iters_v, float seed, float* sink, unsigned long long* cycles) {
    extern __shared__ float lds[];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const int wave = threadIdx.x >> 6;
    f32x4 a[4] = {{seed, 0, 0, 0}, {seed, 0, 0, 0}, {seed, 0, 0, 0}, {seed, 0, 0, 0}};
    f32x2 p[8]; float s[8]; uint32_t u[8]; f32x4 l[8];
    for (int k = 0; k < 8; ++k) { p[k] = f32x2{seed + k, seed - k}; s[k] = seed * 1e-3f + k; u[k] = threadIdx.x + k; l[k] = f32x4{0, 0, 0, 0}; }
    const float* lp = lds + 4 * (threadIdx.x & 255);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if constexpr (MFMA)
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int m = 0; m < 4; ++m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a[m]) : "v"(seed), "v"(1.0f));
    } else {
        for (int i = 0; i < iters_v; ++i) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int k = 0; k < 8; ++k) valu<KIND>(p[k], s[k], u[k], l[k], lp);
            if constexpr (KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = a[0][0] + a[1][1] + a[2][2] + a[3][3];
    for (int k = 0; k < 8; ++k) r += p[k].x + p[k].y + s[k] + (float)u[k] + l[k][0];
    if (r == 12345.678f) sink[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) atomicMax(&cycles[wave < 4 ? 0 : 1], t1 - t0);
}

static const char* kNames[] = {"v_pk_fma_f32", "v_fma_f32", "v_xor_b32", "v_mul_lo_u32", "v_exp_f32", "v_cndmask_b32", "ds_read_b128", "v_mov_b32_dpp"};

template <int KIND, int N, bool MFMA>
static double one(int iters) {
    float* sink; unsigned long long* cyc;
    hipMalloc(&sink, 4096); hipMalloc(&cyc, 16);
    hipFuncSetAttribute((const void*)shadow_kernel<KIND, N, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    shadow_kernel<KIND, N, MFMA><<<256, 256, 100 * 1024>>>(iters, 1.0f, sink, cyc);
    hipMemset(cyc, 0, 16);
    shadow_kernel<KIND, N, MFMA><<<256, 256, 100 * 1024>>>(iters, 1.0f, sink, cyc);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    hipFree(sink); hipFree(cyc);
    return (double)h / iters / 4;          // cycles per group (one MFMA + N instructions)
}

template <int KIND>
static void same_wave(int it) {
    printf("same wave, MFMA + N x %-14s  N=0 %5.1f | alone: N=2 %5.1f N=4 %5.1f N=8 %5.1f | with MFMA: N=2 %5.1f N=4 %5.1f N=8 %5.1f\n", kNames[KIND],
           one<KIND, 0, true>(it), one<KIND, 2, false>(it), one<KIND, 4, false>(it), one<KIND, 8, false>(it),
           one<KIND, 2, true>(it), one<KIND, 4, true>(it), one<KIND, 8, true>(it));
}

template <int KIND>
static void cross(int it) {
    float* sink; unsigned long long* cyc;
    hipMalloc(&sink, 4096); hipMalloc(&cyc, 16);
    unsigned long long h[2];
    auto launch = [&](auto kern, int iv) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        kern<<<256, 512, 100 * 1024>>>(it, iv, 1.0f, sink, cyc);
        hipMemset(cyc, 0, 16);
        kern<<<256, 512, 100 * 1024>>>(it, iv, 1.0f, sink, cyc);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    };
    launch(cross_kernel<KIND, false>, it);
    const double alone = (double)h[1] / it / 32;                 // cycles per instruction, VALU wave alone
    int iv = (int)(it * 128.0 / (alone * 32));                   // size the VALU stream to the MFMA stream's duration
    if (iv < 1) iv = 1;
    launch(cross_kernel<KIND, true>, iv);
    printf("other wave, MFMA | %-14s  alone %5.2f cyc/instr;  together: MFMA wave %5.1f cyc/MFMA, other wave %5.2f cyc/instr (stream sized to the MFMA wave)\n",
           kNames[KIND], alone, (double)h[0] / it / 4, (double)h[1] / iv / 32);
    hipFree(sink); hipFree(cyc);
}

int main() {
    const int it = 20000;
    same_wave<0>(it); same_wave<1>(it); same_wave<2>(it); same_wave<3>(it); same_wave<4>(it); same_wave<5>(it); same_wave<6>(it); same_wave<7>(it);
    cross<0>(it); cross<1>(it); cross<2>(it); cross<3>(it); cross<4>(it); cross<5>(it); cross<6>(it); cross<7>(it);
    return 0;
}
