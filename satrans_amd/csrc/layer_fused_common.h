// Device helpers shared by the register-chained MFMA layer kernels (layer_fused.hip): MFMA chains with
// tokens on the N side, LayerNorm on D-layout fragments, the persistent work distribution, dropout keep bits, LDS image
// staging.  See layer_fused.hip for the layout conventions.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "rng.h"

namespace satrans {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// Diagnostic build only (-DSATRANS_STAMPS): per-phase cycle totals of wave 0 of every workgroup, summed with atomics
// into a module-level array that satrans_debug_read_stamps copies out.  Never compiled into the shipped library.
#ifdef SATRANS_STAMPS
static __device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#define STAMP(slot)                                                                  \
    do {                                                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                \
        if (threadIdx.x == 0) atomicAdd(&g_stamps[slot], now_ - st_prev);            \
        st_prev = now_;                                                              \
    } while (0)
#else
#define STAMP_DECL
#define STAMP(slot)
#endif

constexpr int kFusedBlock = 256;
constexpr int kFusedWaves = kFusedBlock / 64;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- head-dimension rows as packed pairs: one v_pk_fma_f32 does two of the d multiply-adds of a row ---------------
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int d>
__device__ __forceinline__ void load_row(const float* __restrict__ p, f32x2 (&r)[d / 2]) {
#pragma unroll
    for (int e = 0; e < d; e += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + e);
        r[e / 2] = f32x2{t.x, t.y};
        r[e / 2 + 1] = f32x2{t.z, t.w};
    }
}
template <int d>
__device__ __forceinline__ void store_row(float* __restrict__ p, const f32x2 (&r)[d / 2], float s) {
#pragma unroll
    for (int e = 0; e < d; e += 4)
        *reinterpret_cast<float4*>(p + e) =
            make_float4(r[e / 2].x * s, r[e / 2].y * s, r[e / 2 + 1].x * s, r[e / 2 + 1].y * s);
}
template <int d>
__device__ __forceinline__ float dot_row(const f32x2 (&a)[d / 2], const f32x2 (&b)[d / 2]) {
    f32x2 t = a[0] * b[0];
#pragma unroll
    for (int e = 1; e < d / 2; ++e) t = __builtin_elementwise_fma(a[e], b[e], t);
    return t.x + t.y;
}
template <int d>
__device__ __forceinline__ void axpy_row(float s, const f32x2 (&x)[d / 2], f32x2 (&acc)[d / 2]) {
    const f32x2 ss = {s, s};
#pragma unroll
    for (int e = 0; e < d / 2; ++e) acc[e] = __builtin_elementwise_fma(ss, x[e], acc[e]);
}
constexpr float kLog2e = 1.4426950408889634f;
constexpr int kRowChunks = 8;   // the register-resident score row holds 4 * kRowChunks = 32 keys

// out[mt] (16 output features x 16 tokens, D-layout) = sum over KT_*16 input features.
// w: LDS image [K][LDW], w[k*LDW + o] = weight from input feature k to output feature o.
// `wl` = w + 4*g*LDW + n (per-lane base), so every A fragment is one ds_read_b32 at a compile-time offset.
template <int KT_, int MT_, int LDW>
__device__ __forceinline__ void chain(const float* __restrict__ wl, const float (&in)[KT_][4], float (&out)[MT_][4]) {
    constexpr int NS = 4 * KT_;   // contraction steps of 4 input features
    f32x4 acc[MT_];
#pragma unroll
    for (int mt = 0; mt < MT_; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Left alone, hipcc funnels every A fragment through one register pair: ds_read2 -> s_waitcnt lgkmcnt(0) -> 2 MFMAs,
    // i.e. one full LDS round trip per MFMA pair, which at one or two waves per SIMD idles the matrix pipe half of
    // the time.  All A fragments of the chain are therefore read first (NS*MT_ registers) and a scheduling barrier
    // keeps the reads above the MFMAs: the LDS latency is paid once per chain, the MFMAs then issue back to back
    // behind counted lgkmcnt waits.
    float a[NS][MT_];
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int mt = 0; mt < MT_; ++mt) a[st][mt] = wl[(16 * (st >> 2) + (st & 3)) * LDW + 16 * mt];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int mt = 0; mt < MT_; ++mt) acc[mt] = mfma4(a[st][mt], in[st >> 2][st & 3], acc[mt]);
#pragma unroll
    for (int mt = 0; mt < MT_; ++mt) {
        out[mt][0] = acc[mt][0]; out[mt][1] = acc[mt][1]; out[mt][2] = acc[mt][2]; out[mt][3] = acc[mt][3];
    }
}

// The same product through the TRANSPOSE of an image: out[16 mt + i] = sum_k img[(16 mt + i) * LDW + k] * in[k], i.e. the
// contraction runs along the image's rows.  `wl` = img + n * LDW + 4 * g.  The reads of one instruction hit every bank four
// times (row stride = 4 mod 32): 4x the LDS cycles of `chain`'s reads, which is why the backward keeps transposed copies when
// LDS allows and uses this form only when they do not fit (separate Q/K generated weights, flag 'pos').
template <int KT_, int MT_, int LDW>
__device__ __forceinline__ void chain_t(const float* __restrict__ wl, const float (&in)[KT_][4], float (&out)[MT_][4]) {
    constexpr int NS = 4 * KT_;
    f32x4 acc[MT_];
#pragma unroll
    for (int mt = 0; mt < MT_; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the four contraction steps of one 16-feature input tile are four CONSECUTIVE floats of an image row: one 16-byte read
    // instead of four 4-byte reads that hit every bank four times
    float a[NS][MT_];
#pragma unroll
    for (int q = 0; q < KT_; ++q)
#pragma unroll
        for (int mt = 0; mt < MT_; ++mt) {
            const float4 v = *reinterpret_cast<const float4*>(wl + 16 * mt * LDW + 16 * q);
            a[4 * q][mt] = v.x; a[4 * q + 1][mt] = v.y; a[4 * q + 2][mt] = v.z; a[4 * q + 3][mt] = v.w;
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int mt = 0; mt < MT_; ++mt) acc[mt] = mfma4(a[st][mt], in[st >> 2][st & 3], acc[mt]);
#pragma unroll
    for (int mt = 0; mt < MT_; ++mt) {
        out[mt][0] = acc[mt][0]; out[mt][1] = acc[mt][1]; out[mt][2] = acc[mt][2]; out[mt][3] = acc[mt][3];
    }
}

// ---- conflict-free operand paths of the backward kernel (PMC of round 2: SQ_LDS_BANK_CONFLICT 62 % of its LDS-active cycles) ----
// Weight images carry an XOR swizzle: the 16-byte chunk c of row k is stored at chunk c ^ [4 <= k mod 16 < 12].  A by-rows
// 16-byte read of a transposed product (chain_t: lane (n, g) reads row n, chunk g) is serviced in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): rows 4..11 of a group come from the NEXT g, and
// with row stride 4 mod 32 floats they used to land on the banks of rows 12..3 of this g (2-way conflict on 7 of 8 chunks); the
// flip makes g ^ flip(n) constant over a group, i.e. 16 different 4-bank slots.  A by-columns 4-byte read (chain: lane (n, g)
// reads row 4g + r, column n) sees rows whose flip depends on g only, so it stays a permutation inside one aligned 16-float
// segment: conflict-free as before.
__device__ __forceinline__ int img_flip(int row) { return ((row & 15) >= 4 && (row & 15) < 12) ? 4 : 0; }

// global [R][C] -> swizzled LDS image with row stride ld (or its transpose), 16 bytes per load
__device__ __forceinline__ void stage_image_sw(const float* __restrict__ g, float* __restrict__ s, int R, int C, int ld,
                                               bool transpose) {
    const int c4n = C >> 2;
    for (int i = threadIdx.x; i < R * c4n; i += blockDim.x) {
        const int r = i / c4n, c = (i - r * c4n) << 2;
        const float4 v = *reinterpret_cast<const float4*>(g + (size_t)r * C + c);
        if (transpose) {   // image row = source column
            s[c * ld + (r ^ img_flip(c))] = v.x; s[(c + 1) * ld + (r ^ img_flip(c + 1))] = v.y;
            s[(c + 2) * ld + (r ^ img_flip(c + 2))] = v.z; s[(c + 3) * ld + (r ^ img_flip(c + 3))] = v.w;
        } else {
            *reinterpret_cast<float4*>(s + r * ld + (c ^ img_flip(r))) = v;
        }
    }
}

// Token-contraction product with conflict-free operand reads.  acc[MOFF+mt][NOFF+nt] += sum over the wave's 16 token rows of
// A[tok][16mt + .] * G[tok][16nt + .].  al / gl: per-lane bases  buffer + (tile_row0 + 4 g) * ld + n ; step ks adds ONE row.
// Which four rows form a contraction step is free (both operands use the same rows): lane group g takes row 4 g + ks, so the
// two groups of a 32-lane half read rows FOUR apart - with the row stride = 4 mod 8 that is 16 banks apart (`wgrad` reads rows
// one apart: 12 of 16 banks collide).
template <int MT_, int NT_, int MOFF, int NOFF, int LDA, int LDG, int MFULL, int NFULL>
__device__ __forceinline__ void wgrad_r4(const float* al, const float* gl, f32x4 (&acc)[MFULL][NFULL]) {
    float av[4][MT_], gv[4][NT_];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int mt = 0; mt < MT_; ++mt) av[ks][mt] = al[ks * LDA + 16 * mt];
#pragma unroll
        for (int nt = 0; nt < NT_; ++nt) gv[ks][nt] = gl[ks * LDG + 16 * nt];
    }
#ifndef SATRANS_EXP_WGRAD_NOSB
    // (as in `chain`: without it hipcc sinks every step's operand reads to their products - read, wait for everything, four MFMAs,
    //  eight times over: round 6 found that stream in the dW1 products of phase F)
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT_; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT_; ++nt)
                acc[MOFF + mt][NOFF + nt] = mfma4(av[ks][mt], gv[ks][nt], acc[MOFF + mt][NOFF + nt]);
}

// Workgroup barrier for LDS hand-offs inside a tile loop.  __syncthreads() makes hipcc wait for EVERY outstanding memory
// operation (s_waitcnt vmcnt(0)) in front of s_barrier, which stalls the waves on global loads that are meant to fly across
// the phases (next tile's sample index and input row, dy); only the LDS traffic has to be complete here.  Not for hand-offs
// through global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sum over the D features of each token: registers, then the four lane groups
// (v + the lane 16 away, then + the lane 32 away.  gfx950's v_permlane16_swap / v_permlane32_swap exchange 16-lane rows / wave
// halves of two registers in the VALU: given the same value twice they return (own, partner) - the sum of the two is the
// __shfl_xor form's bits, addition commutes, without the two dependent trips through the LDS crossbar (ds_bpermute), whose
// latency nothing hides at one wave per SIMD)
__device__ __forceinline__ float token_sum(float v) {
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    u = __float_as_uint(v);
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LayerNorm over features of a D-layout fragment (torch: biased variance, eps = 1e-6 inside the sqrt).
// gam/bet: LDS vectors [D]; g4 = 4*(lane>>4).  Returns mean / rstd through references when asked for.
template <int KT_>
__device__ __forceinline__ void layer_norm_frag(float (&v)[KT_][4], const float* gam, const float* bet, int g4,
                                                float& mean, float& rstd) {
    constexpr float invD = 1.0f / (16 * KT_);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < KT_; ++t) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
    mean = token_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < KT_; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = v[t][r] - mean;
            q = fmaf(e, e, q);
        }
    rstd = 1.0f / sqrtf(token_sum(q) * invD + 1e-6f);
#pragma unroll
    for (int t = 0; t < KT_; ++t) {
        const float4 gg = *reinterpret_cast<const float4*>(gam + 16 * t + g4);
        const float4 bb = *reinterpret_cast<const float4*>(bet + 16 * t + g4);
        v[t][0] = (v[t][0] - mean) * rstd * gg.x + bb.x;
        v[t][1] = (v[t][1] - mean) * rstd * gg.y + bb.y;
        v[t][2] = (v[t][2] - mean) * rstd * gg.z + bb.z;
        v[t][3] = (v[t][3] - mean) * rstd * gg.w + bb.w;
    }
}

// Work distribution.  The (scenario, tile) pairs of a batch form one global list (scenario-major); a flat grid of
// G workgroups - sized to the number of CUs, so that exactly one round is resident - splits it into G contiguous,
// equally long ranges.  A workgroup whose range crosses a scenario boundary re-stages the generated MetaNet weights
// there.  Scenario rows without samples cost nothing and skewed scenario sizes (real traffic is skewed) balance.
__device__ __forceinline__ int tiles_of(const int32_t* __restrict__ seg, int s, int T) {
    return (seg[s + 1] - seg[s] + T - 1) / T;
}
struct WorkRange {
    int g0, g1, per, total;
};
__device__ __forceinline__ WorkRange work_range(const int32_t* __restrict__ seg, int S, int T, int G, int w) {
    WorkRange r;
    r.total = 0;
    for (int s = 0; s < S; ++s) r.total += tiles_of(seg, s, T);
    r.per = (r.total + G - 1) / G;
    r.g0 = min(r.total, w * r.per);
    r.g1 = min(r.total, r.g0 + r.per);
    return r;
}

struct FusedDrop {
    bool on;
    float scale;
    uint32_t thresh;
    uint32_t key[4];
};

__device__ __forceinline__ FusedDrop fused_drop(const satrans_layer_desc& a) {
    FusedDrop dc;
    dc.on = (a.flags & SATRANS_TRAIN) && a.drop_p > 0.f;
#ifdef SATRANS_DIAG_NODROP      // diagnostic build only: what the dropout hashing and masking cost (results are then those of p = 0)
    dc.on = false;
#endif
    dc.scale = dc.on ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    dc.thresh = drop_threshold(a.drop_p);
    for (int s = 0; s < 4; ++s) dc.key[s] = drop_site_key(a.seed, a.step, a.layer, s);
    return dc;
}

// keep flags of one token lane at one dropout site: the lane owns features 16t + 4g + r of token f, i.e. one block of four
// consecutive element indices per t; bit 4t + r of the result
template <int KT_>
__device__ __forceinline__ uint32_t token_keep_bits(uint32_t sample_key, int f, int D, int g4, uint32_t thresh) {
    uint32_t bits = 0;
#pragma unroll
    for (int t = 0; t < KT_; ++t) bits |= drop_keep4(sample_key, (uint32_t)(f * D + 16 * t + g4) >> 2, thresh) << (4 * t);
    return bits;
}

// weight images resident in LDS for the whole kernel
template <int D, int U>
struct FwdImages {
    static constexpr int LD = D + 4, LU = U + 4;
    float *wq, *wk, *wv, *woT;   // [D][LD]
    float *w1q, *w1k;            // [D][LU]
    float *w2q, *w2k;            // [U][LD]
    float *lnq_g, *lnq_b, *lnk_g, *lnk_b, *ln_g, *ln_b;  // [D]
};

// global [R][C] -> LDS image with row stride ld (or its transpose), 16 bytes per load; C is a multiple of 4, rows of g and of
// the image are 16-byte aligned (C, ld multiples of 4)
__device__ __forceinline__ void stage_image(const float* __restrict__ g, float* __restrict__ s, int R, int C, int ld,
                                            bool transpose) {
    const int c4n = C >> 2;
    for (int i = threadIdx.x; i < R * c4n; i += blockDim.x) {
        const int r = i / c4n, c = (i - r * c4n) << 2;
        const float4 v = *reinterpret_cast<const float4*>(g + (size_t)r * C + c);
        if (transpose) {
            s[c * ld + r] = v.x; s[(c + 1) * ld + r] = v.y; s[(c + 2) * ld + r] = v.z; s[(c + 3) * ld + r] = v.w;
        } else {
            *reinterpret_cast<float4*>(s + r * ld + c) = v;
        }
    }
}

// Several fp32 images at once (stage_image / stage_image_sw with ALL global loads first, then the LDS stores: one stage_image call
// is a loop of load -> wait -> store, eight of them in a row are eight memory latencies with one wave per SIMD and nothing to hide
// them - measured: the backward's prologue and per-scenario staging were ~5 % of the kernel).  SW: the swizzled layout of the backward.  UPT >= ceil(R C / 4 / blockDim) of the largest image.
struct ImageJob {
    const float* g;
    float* s;
    int R, C, ld;
    bool transpose;
};
template <int NJ, int UPT, bool SW>
__device__ __forceinline__ void stage_image_batch(const ImageJob (&jobs)[NJ]) {
    float4 v[NJ][UPT];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int p = 0; p < UPT; ++p) {
            const int i = (int)threadIdx.x + p * (int)blockDim.x;
            if (i < jobs[j].R * (jobs[j].C >> 2)) v[j][p] = *reinterpret_cast<const float4*>(jobs[j].g + 4 * (size_t)i);
        }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const ImageJob& jb = jobs[j];
        const int c4n = jb.C >> 2;
#pragma unroll
        for (int p = 0; p < UPT; ++p) {
            const int i = (int)threadIdx.x + p * (int)blockDim.x;
            if (i < jb.R * c4n) {
                const int r = i / c4n, c = (i - r * c4n) << 2;
                const float4 t = v[j][p];
                float* s_ = jb.s;
                const int ld = jb.ld;
                if (jb.transpose) {   // image row = source column
                    if constexpr (SW) {
                        s_[c * ld + (r ^ img_flip(c))] = t.x; s_[(c + 1) * ld + (r ^ img_flip(c + 1))] = t.y;
                        s_[(c + 2) * ld + (r ^ img_flip(c + 2))] = t.z; s_[(c + 3) * ld + (r ^ img_flip(c + 3))] = t.w;
                    } else {
                        s_[c * ld + r] = t.x; s_[(c + 1) * ld + r] = t.y; s_[(c + 2) * ld + r] = t.z; s_[(c + 3) * ld + r] = t.w;
                    }
                } else {
                    *reinterpret_cast<float4*>(s_ + r * ld + (SW ? (c ^ img_flip(r)) : c)) = t;
                }
            }
        }
    }
}

// layer_norm_frag that also hands back the normalised rows zh = (v - mean) rstd (v = zh gamma + beta: the same bits as above)
template <int KT_>
__device__ __forceinline__ void layer_norm_frag_z(float (&v)[KT_][4], float (&zh)[KT_][4], const float* gam, const float* bet, int g4,
                                                  float& mean, float& rstd) {
    constexpr float invD = 1.0f / (16 * KT_);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < KT_; ++t) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
    mean = token_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < KT_; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = v[t][r] - mean;
            q = fmaf(e, e, q);
        }
    rstd = 1.0f / sqrtf(token_sum(q) * invD + 1e-6f);
#pragma unroll
    for (int t = 0; t < KT_; ++t) {
        const float4 gg = *reinterpret_cast<const float4*>(gam + 16 * t + g4);
        const float4 bb = *reinterpret_cast<const float4*>(bet + 16 * t + g4);
        const float gm[4] = {gg.x, gg.y, gg.z, gg.w}, bt[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            zh[t][r] = (v[t][r] - mean) * rstd;
            v[t][r] = zh[t][r] * gm[r] + bt[r];
        }
    }
}

// MetaNet of one role on a D-layout fragment: out = LN(drop(relu(in W1) W2) + in)     submodules.py:77-103
// Also hands back the hidden activations, the pre-norm rows and the statistics for the backward pass.
template <int D, int U>
__device__ __forceinline__ void metanet_frag(const float* w1l, const float* w2l, const float* gam, const float* bet,
                                             int g4, const FusedDrop& dc, int site, uint32_t sample_key, int f,
                                             const float (&in)[D / 16][4], float (&h)[U / 16][4],
                                             float (&out)[D / 16][4], float& mean, float& rstd, float (*zh)[4] = nullptr) {
    constexpr int KT = D / 16, UT = U / 16;
    chain<KT, UT, U + 4>(w1l, in, h);
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[t][r] = fmaxf(h[t][r], 0.f);
    chain<UT, KT, D + 4>(w2l, h, out);
    const uint32_t kb = dc.on ? token_keep_bits<KT>(sample_key, f, D, g4, dc.thresh) : 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = out[t][r];
            if (dc.on) m = (kb >> (4 * t + r)) & 1u ? m * dc.scale : 0.f;
            out[t][r] = m + in[t][r];
        }
    if (zh) layer_norm_frag_z<KT>(out, *reinterpret_cast<float (*)[KT][4]>(zh), gam, bet, g4, mean, rstd);
    else layer_norm_frag<KT>(out, gam, bet, g4, mean, rstd);
}

struct SlabOffF {
    int wq, wk, wv, wo, w1q, w2q, w1k, w2k, ln, lnq, lnk, total;
};
__host__ __device__ inline SlabOffF slab_offsets_f(int D, int U) {   // same layout as layer_lds.hip
    SlabOffF s;
    int o = 0;
    s.wq = o; o += D * D;
    s.wk = o; o += D * D;
    s.wv = o; o += D * D;
    s.wo = o; o += D * D;
    s.w1q = o; o += D * U;
    s.w2q = o; o += U * D;
    s.w1k = o; o += D * U;
    s.w2k = o; o += U * D;
    s.ln = o; o += 2 * D;
    s.lnq = o; o += 2 * D;
    s.lnk = o; o += 2 * D;
    s.total = o;
    return s;
}

template <int KT_>
__device__ __forceinline__ void load_frag(const float* row, float (&v)[KT_][4], bool ok = true) {
#pragma unroll
    for (int t = 0; t < KT_; ++t) {
        const float4 q = *reinterpret_cast<const float4*>(row + 16 * t);
        v[t][0] = ok ? q.x : 0.f; v[t][1] = ok ? q.y : 0.f; v[t][2] = ok ? q.z : 0.f; v[t][3] = ok ? q.w : 0.f;
    }
}
template <int KT_>
__device__ __forceinline__ void store_frag(float* row, const float (&v)[KT_][4], bool ok = true) {
#pragma unroll
    for (int t = 0; t < KT_; ++t)
        *reinterpret_cast<float4*>(row + 16 * t) = ok ? make_float4(v[t][0], v[t][1], v[t][2], v[t][3])
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
}

// acc[MOFF+mt][NOFF+nt] += sum over the wave's 16 token rows of A[tok][16mt + .] * G[tok][16nt + .]   (mt < MT_, nt < NT_)
// al / gl: per-lane bases  buffer + (tile_row0 + g)*ld + n ; step ks adds 4 rows
template <int MT_, int NT_, int MOFF, int NOFF, int LDA, int LDG, int MFULL, int NFULL>
__device__ __forceinline__ void wgrad(const float* al, const float* gl, f32x4 (&acc)[MFULL][NFULL]) {
    // all operands of the four token steps are read first (4*(MT_+NT_) registers), then the MFMAs run back to back
    float av[4][MT_], gv[4][NT_];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int mt = 0; mt < MT_; ++mt) av[ks][mt] = al[4 * ks * LDA + 16 * mt];
#pragma unroll
        for (int nt = 0; nt < NT_; ++nt) gv[ks][nt] = gl[4 * ks * LDG + 16 * nt];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT_; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT_; ++nt)
                acc[MOFF + mt][NOFF + nt] = mfma4(av[ks][mt], gv[ks][nt], acc[MOFF + mt][NOFF + nt]);
}

// LayerNorm forward that keeps the normalised rows and 1/std for the backward pass
template <int KT_>
__device__ __forceinline__ void layer_norm_keep(const float (&z)[KT_][4], float (&zh)[KT_][4], float& rstd) {
    constexpr float invD = 1.0f / (16 * KT_);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < KT_; ++t) s += (z[t][0] + z[t][1]) + (z[t][2] + z[t][3]);
    const float mean = token_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < KT_; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = z[t][r] - mean;
            q = fmaf(e, e, q);
        }
    rstd = 1.0f / sqrtf(token_sum(q) * invD + 1e-6f);
#pragma unroll
    for (int t = 0; t < KT_; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) zh[t][r] = (z[t][r] - mean) * rstd;
}

// LayerNorm backward on a D-layout fragment: g (gradient of the normalised-and-scaled output) becomes the gradient
// of the pre-norm rows; gamma / beta gradients accumulate per lane (reduced over lanes and waves at kernel end).
template <int KT_>
__device__ __forceinline__ void layer_norm_bwd(float (&g)[KT_][4], const float (&zh)[KT_][4], float rstd,
                                               const float* gam, int g4, float (&acc_g)[KT_][4],
                                               float (&acc_b)[KT_][4]) {
    constexpr float invD = 1.0f / (16 * KT_);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int t = 0; t < KT_; ++t) {
        const float4 gm4 = *reinterpret_cast<const float4*>(gam + 16 * t + g4);
        const float gm[4] = {gm4.x, gm4.y, gm4.z, gm4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc_g[t][r] = fmaf(g[t][r], zh[t][r], acc_g[t][r]);
            acc_b[t][r] += g[t][r];
            g[t][r] *= gm[r];
            m1 += g[t][r];
            m2 = fmaf(g[t][r], zh[t][r], m2);
        }
    }
    m1 = token_sum(m1) * invD;
    m2 = token_sum(m2) * invD;
#pragma unroll
    for (int t = 0; t < KT_; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) g[t][r] = rstd * (g[t][r] - m1 - zh[t][r] * m2);
}

static int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

}  // namespace satrans
