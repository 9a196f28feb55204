// Do fp32 MFMAs and plain VALU instructions of DIFFERENT waves on one SIMD overlap?  8 waves per SIMD; in the mixed launches
// waves 0-3 of a block run the MFMA loop and waves 4-7 the VALU loop (4 + 4 per SIMD), against each loop alone with
// the same 4 waves per SIMD.  If the mixed time is the larger of the two, the pipes overlap; if it is their sum, they share issue.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_valu_overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// MODE bit0: MFMA waves active, bit1: VALU waves active; VK: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 ds_read_b128
template <int MODE, int VK>
__global__ __launch_bounds__(512) void overlap_kernel(float* out, int iters, float seed) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = seed;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float a[16];
    f32x2 b[16];
    f32x4 q[4] = {{seed, 0.f, 0.f, 0.f}, {0.f, seed, 0.f, 0.f}, {0.f, 0.f, seed, 0.f}, {0.f, 0.f, 0.f, seed}};
    f32x4 l[4];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; b[i] = f32x2{a[i], a[i] + 0.5f}; }
    const bool mfma_wave = wave < 4;        // (waves are placed round robin over the 4 SIMDs: waves 0-3 and 4-7 each cover all of them)
    if (mfma_wave) {
        if (MODE & 1)
            for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %1, %0" : "+v"(q[i & 3]) : "v"(a[i]));
                REP16(X)
#undef X
            }
    } else {
        if (MODE & 2)
            for (int it = 0; it < iters * 8; ++it) {
                if constexpr (VK == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
                    REP16(X)
#undef X
                } else if constexpr (VK == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(b[i]));
                    REP16(X)
#undef X
                } else if constexpr (VK == 2) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                    REP16(X)
#undef X
                } else {
                    const float* base = lds + (threadIdx.x & 63) * 4;
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l[i & 3]) : "v"((unsigned)(size_t)base), "n"((i) * 1024));
                    REP16(X)
#undef X
                    asm volatile("s_waitcnt lgkmcnt(0)");
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i] + b[i].x + b[i].y + q[i & 3].x + q[i & 3].w + l[i & 3].x;
    if (s == 12345.678f) out[0] = s;
}

template <int MODE, int VK>
static double run(float* out, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    overlap_kernel<MODE, VK><<<1024, 512>>>(out, 16, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    overlap_kernel<MODE, VK><<<1024, 512>>>(out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int VK>
static void report(const char* name, float* out, int iters) {
    const double m = run<1, VK>(out, iters), v = run<2, VK>(out, iters), both = run<3, VK>(out, iters);
    printf("%-14s  MFMA waves alone %7.3f ms   %s waves alone %7.3f ms   both %7.3f ms   (sum %7.3f, max %7.3f)\n", name, m, name, v,
           both, m + v, m > v ? m : v);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 64);
    const int iters = 4000;      // 1024 blocks x 8 waves over 1024 SIMDs: 4 MFMA waves + 4 VALU waves per SIMD
    report<0>("v_fma_f32", out, iters);
    report<1>("v_pk_fma_f32", out, iters);
    report<2>("v_exp_f32", out, iters);
    report<3>("ds_read_b128", out, iters);
    return 0;
}
