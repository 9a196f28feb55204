// Issue cost of the VALU instruction classes the lazy Adam replay is made of, on the device it runs on: every kernel is a loop of
// 16 independent instructions of one class, 8 waves per SIMD, timed with HIP events; the table prints the time per wave-instruction
// relative to v_fma_f32 (= one 4-cycle issue slot).
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x2 = __attribute__((ext_vector_type(2))) float;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
    float a[16];
    f32x2 b[16];
    double d[16];
    const unsigned long long mask = 0x5555aaaa5555aaaaull;
    const int lane4 = ((threadIdx.x + 1) & 63) * 4;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    f32x4 q[4] = {{seed, 0.f, 0.f, 0.f}, {0.f, seed, 0.f, 0.f}, {0.f, 0.f, seed, 0.f}, {0.f, 0.f, 0.f, seed}};
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; b[i] = f32x2{a[i], a[i] + 0.5f}; d[i] = a[i]; }
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(b[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 2) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 3) {
#define X(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 4) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 5) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 6) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 7) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(b[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 8) {
#define X(i) asm volatile("v_min3_f32 %0, %0, %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 9) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 10) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 11) {
#define X(i) asm volatile("v_mov_b32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 12) {
            // mixed: 8 packed fma + 8 rcp, interleaved - do transcendentals overlap with the packed pipe?
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_rcp_f32 %1, %1" : "+v"(b[i]), "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 13) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_mul_f64 %1, %1, %1" : "+v"(b[i]), "+v"(d[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 14) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(b[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 15) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 16) {
#define X(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 17) {
#define X(i) asm volatile("v_min_f32 %0, %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 18) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 19) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 20) {
#define X(i) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 21) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 22) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %0" : : "v"(a[i]) : "vcc");
            REP16(X)
#undef X
        } else if constexpr (KIND == 23) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 24) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(b[(i + 5) & 15]), "v"(b[(i + 9) & 15]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 25) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1" : "+v"(a[i]), "+v"(b[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 26) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_rcp_f32 %1, %1" : "+v"(a[i]), "+v"(b[i].x));
            REP16(X)
#undef X
        } else if constexpr (KIND == 27) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_mul_f64 %1, %1, %1" : "+v"(a[i]), "+v"(d[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 28) {
#define X(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %1, %0" : "+v"(q[i & 3]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 29) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %0, %1" : "+v"(a[i]) : "s"(mask));
            REP16(X)
#undef X
        } else if constexpr (KIND == 30) {
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 31) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %1, %2, vcc" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]) : "vcc");
            REP16(X)
#undef X
        } else if constexpr (KIND == 32) {
#define X(i) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "=v"(a[i]) : "v"(a[(i + 5) & 15]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 33) {
#define X(i) asm volatile("v_max_f32 %0, 0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 34) {
#define X(i) asm volatile("v_and_b32 %0, %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 35) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 36) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 37) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 38) {
#define X(i) asm volatile("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 39) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 40) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(lane4));
            REP16(X)
#undef X
        } else if constexpr (KIND == 41) {
#define X(i) asm volatile("ds_swizzle_b32 %0, %0 offset:0x041F\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 42) {
#define X(i) asm volatile("v_readlane_b32 s20, %0, 3\n\ts_nop 3\n\tv_mov_b32 %0, s20" : "+v"(a[i]) : : "s20");
            REP16(X)
#undef X
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i] + b[i].x + b[i].y + (float)d[i] + q[i & 3].x + q[i & 3].w;
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
static double run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<KIND><<<2048, 256>>>(out, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<KIND><<<2048, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 64);
    const int iters = 20000;
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_rcp_f32", "v_rsq_f32", "v_cvt_f64_f32", "v_mul_f64", "v_cvt_f32_f64",
                           "v_pk_mul_f32", "v_min3_f32", "v_sqrt_f32", "v_fma_f64", "v_mov_b32", "pk_fma + rcp (pair)",
                           "pk_fma + mul_f64 (pair)", "v_pk_add_f32",
                           "v_mul_f32", "v_add_f32", "v_min_f32", "v_max3_f32", "v_cndmask_b32", "v_add_u32", "v_exp_f32", "v_cmp_lt_f32 (vcc)", "v_fma_f32 3 distinct srcs", "v_pk_fma_f32 3 distinct srcs", "v_fma_f32 + v_pk_fma_f32 (pair)", "v_fma_f32 + v_rcp_f32 (pair)", "v_fma_f32 + v_mul_f64 (pair)", "v_mfma_f32_16x16x4_f32",
                           "v_cndmask_b32 e64, sgpr mask", "v_cndmask_b32 distinct srcs, vcc", "v_cmp + v_cndmask (pair)", "v_cndmask_b32 0, v, vcc", "v_max_f32 v, 0", "v_and_b32", "v_lshlrev_b32", "v_mul_lo_u32", "v_fmac_f32", "v_mul_f32 dpp row_shr:1", "v_mov_b32 dpp row_shr:1", "ds_bpermute_b32", "ds_swizzle_b32", "v_readlane_b32 + v_mov from sgpr"};
    double ms[43];
    ms[0] = run<0>(out, iters); ms[1] = run<1>(out, iters); ms[2] = run<2>(out, iters); ms[3] = run<3>(out, iters);
    ms[4] = run<4>(out, iters); ms[5] = run<5>(out, iters); ms[6] = run<6>(out, iters); ms[7] = run<7>(out, iters);
    ms[8] = run<8>(out, iters); ms[9] = run<9>(out, iters); ms[10] = run<10>(out, iters); ms[11] = run<11>(out, iters);
    ms[12] = run<12>(out, iters); ms[13] = run<13>(out, iters); ms[14] = run<14>(out, iters);
    ms[15] = run<15>(out, iters); ms[16] = run<16>(out, iters); ms[17] = run<17>(out, iters); ms[18] = run<18>(out, iters); ms[19] = run<19>(out, iters); ms[20] = run<20>(out, iters); ms[21] = run<21>(out, iters); ms[22] = run<22>(out, iters); ms[23] = run<23>(out, iters); ms[24] = run<24>(out, iters); ms[25] = run<25>(out, iters); ms[26] = run<26>(out, iters); ms[27] = run<27>(out, iters); ms[28] = run<28>(out, iters);
    ms[29] = run<29>(out, iters); ms[30] = run<30>(out, iters); ms[31] = run<31>(out, iters); ms[32] = run<32>(out, iters); ms[33] = run<33>(out, iters); ms[34] = run<34>(out, iters); ms[35] = run<35>(out, iters); ms[36] = run<36>(out, iters); ms[37] = run<37>(out, iters); ms[38] = run<38>(out, iters); ms[39] = run<39>(out, iters); ms[40] = run<40>(out, iters); ms[41] = run<41>(out, iters); ms[42] = run<42>(out, iters);
    // 2048 blocks x 4 waves over 256 CUs x 4 SIMDs = 8 waves per SIMD, each issuing iters x 16 instructions (x 2 for the pairs)
    for (int k = 0; k < 43; ++k) {
        const double per = ms[k] * 1e6 / ((double)iters * 16 * 8);          // ns per wave-instruction (or pair) per SIMD
        printf("%-34s %8.3f ms  %7.3f ns per issue  %5.2f x v_fma_f32\n", names[k], ms[k], per, ms[k] / ms[0]);
    }
    return 0;
}
