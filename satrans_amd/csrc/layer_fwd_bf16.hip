// Meta_Transformer_Layer EVALUATION forward with the dense products on the bf16 matrix pipe (BASELINE.json configs[1]:
// "bf16 forward, fp32 ref-parity check").  Reference: models/satrans.py:50-100, models/submodules.py:77-103.
//
// Same decomposition as layer_fwd_fused_kernel (layer_fused.hip): persistent workgroups over the (scenario, tile) list,
// every weight staged into LDS once, the token-wise chain  x -> {q0,k0,v} -> relu(q0 W1) -> (.) W2 -> +q0 -> LayerNorm
// per 16-token tile in one wave with tokens on the N side so that an accumulator is the next product's B operand, the
// feature x feature attention as one lane per (sample, head, query row).  What changes:
//
//   * products run on v_mfma_f32_16x16x32_bf16 (16x the f32 MFMA rate, K = 32 per instruction: ONE instruction per 16x16
//     output tile at D = 32 instead of eight), operands rounded to bf16 (RNE, v_cvt_pk_bf16_f32), accumulation in fp32;
//   * LayerNorm, residuals, softmax and the attention dot products stay in fp32 (they are VALU work either way);
//   * weights are converted to bf16 ONCE per workgroup while they are staged into LDS, in an image whose rows are already in
//     the order the B operand of a chained product presents its contraction index: lane group g, element j of a K-step holds
//     input feature 16 (j / 4) + 4 g + j % 4 (the C-layout of the previous product), so an A fragment is one 16-byte read.
//
// Evaluation only (no dropout, no attention capture); the fp32 kernels remain the training path and the parity reference.
// Logit error against the fp32 oracle is ~1e-2 (8.1e-3 all-bf16 near init) and is reported separately.
#include "layer_fused_common.h"

namespace satrans {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// position of contraction index k inside an image row: K-step s = k / 32, then [g][j] with k % 32 = 16 (j / 4) + 4 g + j % 4
__device__ __forceinline__ int bf16_pos(int k) {
    const int s = k >> 5, kk = k & 31;
    const int t = kk >> 4, g = (kk & 15) >> 2, r = kk & 3;
    return 32 * s + 8 * g + 4 * t + r;
}

// global fp32 weight -> bf16 image [OUT][RS] (RS = K + 8: 16 bytes of padding keep the 16-byte fragment reads of 16
// consecutive rows on distinct banks).  in_major: g[k * OUT + o] (W[in][out]: W_Query/Key/Value and the generated W1 [D][U],
// W2 [U][D]); else g[o * K + k] (nn.Linear weight [out][in]: Out_linear).
__device__ __forceinline__ void stage_bf16(const float* __restrict__ g, __bf16* __restrict__ img, int OUT, int K, int RS,
                                           bool in_major) {
    for (int i = threadIdx.x; i < OUT * K; i += blockDim.x) {
        int o, k;
        if (in_major) { k = i / OUT; o = i - k * OUT; } else { o = i / K; k = i - o * K; }
        img[o * RS + bf16_pos(k)] = (__bf16)g[i];
    }
}

// out[mt] (16 output features x 16 tokens, D-layout) = sum over the 16 * KT_ input features of `in` (D-layout, fp32)
// il: image + n * RS + 8 g (this lane's row inside an output tile and its lane group's 8 elements of a K-step)
template <int KT_, int MT_, int RS>
__device__ __forceinline__ void chain_bf16(const __bf16* __restrict__ il, const float (&in)[KT_][4], float (&out)[MT_][4]) {
    static_assert(KT_ % 2 == 0, "a K-step of the bf16 MFMA spans two 16-feature tiles");
    f32x4 acc[MT_];
#pragma unroll
    for (int mt = 0; mt < MT_; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KT_ / 2; ++s) {
        bf16x8 b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = (__bf16)in[2 * s + (j >> 2)][j & 3];
#pragma unroll
        for (int mt = 0; mt < MT_; ++mt) {
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(il + 16 * mt * RS + 32 * s);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b, acc[mt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT_; ++mt) {
        out[mt][0] = acc[mt][0]; out[mt][1] = acc[mt][1]; out[mt][2] = acc[mt][2]; out[mt][3] = acc[mt][3];
    }
}

// FT: the field count as a constant (0 = a.F).  With it and d = 8 the attention runs on the matrix pipe as in the fused backward
// (layer_fused.hip, phases B / D / E): v_mfma_f32_4x4x1_16b_f32, one lane per (sample, head, query row) in whole groups of
// FP = 4 ceil(F / 4) lanes per wave, S^T = K q from the key rows picked by lane & 3 and the lane's own query row, the softmax
// lane-local on the accumulator registers, o = P V from the value feature pairs picked by lane & 3 - fp32 throughout.  With
// the products on the bf16 pipe the attention is ~2/3 of this kernel, so this is where its time goes.
template <int D, int U, int H, int WAVES, int FT = 0>
__global__ __launch_bounds__(64 * WAVES) void layer_fwd_bf16_kernel(satrans_layer_desc a, int Tsamp, float* __restrict__ y) {
    constexpr int KT = D / 16, UT = U / 16, d = D / H, LD = D + 4, KD = D + 8, KU = U + 8;
    constexpr bool MFA = FT != 0 && d == 8;
    extern __shared__ __align__(16) float lds[];
    const int F = FT ? FT : a.F;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4, g4 = 4 * g;
    const bool meta_q = a.flags & SATRANS_META_Q, meta_k = a.flags & SATRANS_META_K;
    const bool same_tab = a.tab_q == a.tab_k;
    const bool relu_out = a.flags & SATRANS_RELU_OUT, use_res = !(a.flags & SATRANS_NO_RES);

    // ---- carve LDS: bf16 images first (16-byte aligned pieces), then fp32 vectors and the q / k / v row buffers ---------
    __bf16* bp = reinterpret_cast<__bf16*>(lds);
    auto take_b = [&](int cnt) { __bf16* r = bp; bp += (cnt + 7) & ~7; return r; };
    __bf16* wq = take_b(D * KD); __bf16* wk = take_b(D * KD); __bf16* wv = take_b(D * KD); __bf16* wo = take_b(D * KD);
    __bf16* w1q = take_b(U * KD); __bf16* w2q = take_b(D * KU);
    __bf16* w1k = same_tab ? w1q : take_b(U * KD);
    __bf16* w2k = same_tab ? w2q : take_b(D * KU);
    float* p = reinterpret_cast<float*>(bp);
    auto take = [&](int cnt) { float* r = p; p += (cnt + 3) & ~3; return r; };
    float* lnq_g = take(D); float* lnq_b = take(D); float* lnk_g = take(D); float* lnk_b = take(D);
    float* ln_g = take(D); float* ln_b = take(D);
    const int rows = ((Tsamp * F + 15) >> 4) << 4;
    float* sq = take(rows * LD);
    float* sk = take(rows * LD);
    float* sv = take(rows * LD);

    stage_bf16(a.w_query, wq, D, D, KD, true);
    stage_bf16(a.w_key, wk, D, D, KD, true);
    stage_bf16(a.w_value, wv, D, D, KD, true);
    stage_bf16(a.w_out, wo, D, D, KD, false);
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        ln_g[i] = a.ln_g[i]; ln_b[i] = a.ln_b[i];
        if (meta_q) { lnq_g[i] = a.lnq_g[i]; lnq_b[i] = a.lnq_b[i]; }
        if (meta_k) { lnk_g[i] = a.lnk_g[i]; lnk_b[i] = a.lnk_b[i]; }
    }
    const int il_d = n * KD + 8 * g, il_u = n * KU + 8 * g;      // this lane's fragment inside an image with rows of D / U inputs
    const float inv_sqrt_d = 1.0f / sqrtf((float)d);
    const WorkRange wr = work_range(a.seg, a.S, Tsamp, gridDim.x, blockIdx.x);

    int pre = 0;
    for (int scen = 0; scen < a.S && pre < wr.g1; ++scen) {
      const int nt_s = tiles_of(a.seg, scen, Tsamp);
      const int t0 = max(wr.g0, pre) - pre, t1 = min(wr.g1, pre + nt_s) - pre;
      pre += nt_s;
      if (t0 >= t1) continue;
      // ---- this scenario's generated MetaNet weights (the previous tile loop ended on a barrier) ------------------
      if (meta_q) {
          const float* row = a.tab_q + (size_t)scen * a.tab_stride;
          stage_bf16(row, w1q, U, D, KD, true);               // W1 [D][U]: in-major, K = D
          stage_bf16(row + D * U, w2q, D, U, KU, true);       // W2 [U][D]: in-major, K = U
      }
      if (meta_k && (!same_tab || !meta_q)) {
          const float* row = a.tab_k + (size_t)scen * a.tab_stride;
          stage_bf16(row, w1k, U, D, KD, true);
          stage_bf16(row + D * U, w2k, D, U, KU, true);
      }
      __syncthreads();
      const int lo = a.seg[scen], hi = a.seg[scen + 1];
      for (int tile = t0; tile < t1; ++tile) {
        const int first = lo + tile * Tsamp;
        const int32_t* samp = a.order + first;
        const int nS = min(Tsamp, hi - first), ntok = nS * F, ntt = (ntok + 15) >> 4;

        // ---- phase 1: projections + MetaNet per 16-token tile, all in registers --------------------------
        for (int tt = wave; tt < ntt; tt += WAVES) {
            const int tok = 16 * tt + n;
            const bool valid = tok < ntok;
            const int ls = valid ? tok / F : 0, f = valid ? tok - ls * F : 0;
            const int b = samp[ls];
            float x[KT][4], q[KT][4], k[KT][4], v[KT][4];
            load_frag<KT>(layer_x_row(a, b, f, F, D) + g4, x);
            chain_bf16<KT, KT, KD>(wq + il_d, x, q);                                     // satrans.py:55-57
            chain_bf16<KT, KT, KD>(wk + il_d, x, k);
            chain_bf16<KT, KT, KD>(wv + il_d, x, v);
            auto metanet = [&](float (&z)[KT][4], const __bf16* w1, const __bf16* w2, const float* gam, const float* bet) {
                float h[UT][4], o[KT][4];                                                // submodules.py:77-103
                chain_bf16<KT, UT, KD>(w1 + il_d, z, h);
#pragma unroll
                for (int t = 0; t < UT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[t][r] = fmaxf(h[t][r], 0.f);
                chain_bf16<UT, KT, KU>(w2 + il_u, h, o);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[t][r] += o[t][r];
                float mean, rstd;
                layer_norm_frag<KT>(z, gam, bet, g4, mean, rstd);
            };
            if (meta_q) metanet(q, w1q, w2q, lnq_g, lnq_b);                              // satrans.py:60-66
            if (meta_k) metanet(k, w1k, w2k, lnk_g, lnk_b);                              // satrans.py:67-73
            store_frag<KT>(sq + (size_t)tok * LD + g4, q);
            store_frag<KT>(sk + (size_t)tok * LD + g4, k);
            store_frag<KT>(sv + (size_t)tok * LD + g4, v);
        }
        __syncthreads();

        // ---- phase 2: attention in fp32, one lane per (sample, head, query row)  (satrans.py:75-90) -------------------
        // two passes over the keys in chunks of four (maximum of the scaled scores, then exp2 / sum / PV with the scores
        // recomputed); padding keys of the last chunk read the last real row and are masked
        if constexpr (MFA) {
          constexpr int FP = (FT + 3) & ~3, NJB = FP / 4, SHW = 64 / FP;      // SHW whole (sample, head) groups per wave
          const int grp = lane / FP, m_i = lane - grp * FP, sub = lane & 3;   // (FP is a multiple of 4: m_i & 3 = lane & 3)
          const float sc_scale = inv_sqrt_d * kLog2e;
          const int ngrp = nS * H;
          for (int g0 = wave * SHW; g0 < ngrp; g0 += WAVES * SHW) {           // (wave-uniform: every lane runs the products)
            // spare lanes of the wave and groups beyond the tile repeat a real group: same operands, nothing stored
            const int sh = min(g0 + min(grp, SHW - 1), ngrp - 1);
            const bool own = grp < SHW && g0 + grp < ngrp && m_i < FT;
            const int ls = sh / H, h = sh - ls * H, iq = min(m_i, FT - 1);
            float* qrow = sq + (size_t)(ls * FT + iq) * LD + h * d;
            float qe[d];
            {
                const float4 q0_ = *reinterpret_cast<const float4*>(qrow), q1_ = *reinterpret_cast<const float4*>(qrow + 4);
                qe[0] = q0_.x; qe[1] = q0_.y; qe[2] = q0_.z; qe[3] = q0_.w; qe[4] = q1_.x; qe[5] = q1_.y; qe[6] = q1_.z; qe[7] = q1_.w;
            }
            float ka[NJB][d];
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) {      // (a padding key of the last block reads a row of the buffer: masked below)
                const float* kr = sk + (size_t)min(ls * FT + 4 * jb + sub, rows - 1) * LD + h * d;
                const float4 k0_ = *reinterpret_cast<const float4*>(kr), k1_ = *reinterpret_cast<const float4*>(kr + 4);
                ka[jb][0] = k0_.x; ka[jb][1] = k0_.y; ka[jb][2] = k0_.z; ka[jb][3] = k0_.w;
                ka[jb][4] = k1_.x; ka[jb][5] = k1_.y; ka[jb][6] = k1_.z; ka[jb][7] = k1_.w;
            }
            f32x2 vp[FT];
            {
                const float* vb = sv + (size_t)(ls * FT) * LD + h * d + 2 * sub;
#pragma unroll
                for (int j = 0; j < FT; ++j) vp[j] = *reinterpret_cast<const f32x2*>(vb + (size_t)j * LD);
            }
            f32x4 sc4[NJB];
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) sc4[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < d; ++e)
#pragma unroll
                for (int jb = 0; jb < NJB; ++jb)
                    sc4[jb] = __builtin_amdgcn_mfma_f32_4x4x1f32(ka[jb][e], qe[e], sc4[jb], 0, 0, 0);
            float ex[FT];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < FT; ++j) {
                ex[j] = sc4[j >> 2][j & 3] * sc_scale;
                mx = fmaxf(mx, ex[j]);
            }
            float sum = 0.f;
            f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < FT; ++j) {
                ex[j] = __builtin_amdgcn_exp2f(ex[j] - mx);
                sum += ex[j];
                oa = __builtin_amdgcn_mfma_f32_4x4x1f32(vp[j].x, ex[j], oa, 0, 0, 0);
                ob = __builtin_amdgcn_mfma_f32_4x4x1f32(vp[j].y, ex[j], ob, 0, 0, 0);
            }
            // the attention output takes the place of this lane's own q row slice (nobody else reads it)
            const float inv = 1.0f / sum;
            if (own) {
                *reinterpret_cast<float4*>(qrow) = make_float4(oa[0] * inv, ob[0] * inv, oa[1] * inv, ob[1] * inv);
                *reinterpret_cast<float4*>(qrow + 4) = make_float4(oa[2] * inv, ob[2] * inv, oa[3] * inv, ob[3] * inv);
            }
          }
        } else {
          const float sc_scale = inv_sqrt_d * kLog2e;
          for (int task = threadIdx.x; task < nS * H * F; task += 64 * WAVES) {
            const int ls = task / (H * F), rem = task - ls * H * F;
            const int h = rem / F, i = rem - h * F;
            float* qrow = sq + (size_t)(ls * F + i) * LD + h * d;
            const float* kbase = sk + (size_t)(ls * F) * LD + h * d;
            const float* vbase = sv + (size_t)(ls * F) * LD + h * d;
            f32x2 qi[d / 2];
            load_row<d>(qrow, qi);
            float mx = -INFINITY;
#pragma unroll 1
            for (int j0 = 0; j0 < F; j0 += 4) {
                f32x2 kr[4][d / 2];
#pragma unroll
                for (int u = 0; u < 4; ++u) load_row<d>(kbase + (size_t)min(j0 + u, F - 1) * LD, kr[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) mx = fmaxf(mx, dot_row<d>(qi, kr[u]) * sc_scale);
            }
            f32x2 oacc[d / 2];
#pragma unroll
            for (int e = 0; e < d / 2; ++e) oacc[e] = f32x2{0.f, 0.f};
            float sum = 0.f;
#pragma unroll 1
            for (int j0 = 0; j0 < F; j0 += 4) {
                f32x2 kr[4][d / 2], vr[4][d / 2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    load_row<d>(kbase + (size_t)min(j0 + u, F - 1) * LD, kr[u]);
                    load_row<d>(vbase + (size_t)min(j0 + u, F - 1) * LD, vr[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float ex = j0 + u < F ? __builtin_amdgcn_exp2f(dot_row<d>(qi, kr[u]) * sc_scale - mx) : 0.f;
                    sum += ex;
                    axpy_row<d>(ex, vr[u], oacc);
                }
            }
            // the attention output takes the place of this task's own q row (nobody else reads it)
            store_row<d>(qrow, oacc, 1.0f / sum);
          }
        }
        __syncthreads();

        // ---- phase 3: Out_linear, residual, LayerNorm per 16-token tile (satrans.py:91-99) ------------
        for (int tt = wave; tt < ntt; tt += WAVES) {
            const int tok = 16 * tt + n;
            const bool valid = tok < ntok;
            const int ls = valid ? tok / F : 0, f = valid ? tok - ls * F : 0;
            const int b = samp[ls];
            float o[KT][4], u[KT][4], x[KT][4];
            load_frag<KT>(sq + (size_t)tok * LD + g4, o);
            chain_bf16<KT, KT, KD>(wo + il_d, o, u);
            load_frag<KT>(layer_x_row(a, b, f, F, D) + g4, x);
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float val = u[t][r];
                    if (relu_out) val = fmaxf(val, 0.f);
                    if (use_res) val += x[t][r];
                    u[t][r] = val;
                }
            float mean, rstd;
            layer_norm_frag<KT>(u, ln_g, ln_b, g4, mean, rstd);
            if (valid) store_frag<KT>(y + ((size_t)b * F + f) * D + g4, u);
        }
        __syncthreads();
      }
    }
}

// The whole stack of NL layers in ONE launch (evaluation: nothing is handed to a backward): a tile's rows stay in LDS between
// the layers - read from HBM once, written once - and every layer's weight images are staged once per workgroup.  Same code
// per layer as layer_fwd_bf16_kernel, i.e. the same bits.
struct Bf16Stack {
    satrans_layer_desc d[4];
    int L;
};
// ... and the head behind it (satrans.py:244-255: flatten + dense columns + Linear + sigmoid; head.hip's head_kernel): a tile holds
// whole samples, so the last layer's rows never leave LDS.  One wave per sample, lane i takes the 16-byte chunks i, i + 64, ... of
// the flattened row in head_kernel's order and the same butterfly sum: the same bits.  prob == nullptr: no head, y is written.
struct Bf16Head {
    const float* w;
    const float* bias;
    const float* dense;
    int64_t dense_stride;
    int32_t dense_col[2];
    int32_t n_dense;
    float* prob;
    float* logit;
};

template <int D, int U, int H, int WAVES, int FT = 0>
__global__ __launch_bounds__(64 * WAVES) void stack_fwd_bf16_kernel(Bf16Stack sa, int Tsamp, float* __restrict__ y, Bf16Head hd) {
    const satrans_layer_desc& a = sa.d[0];      // shape, flags, scenario segments and the input rows: the first layer's
    const int NL = sa.L;
    constexpr int KT = D / 16, UT = U / 16, d = D / H, LD = D + 4, KD = D + 8, KU = U + 8;
    constexpr bool MFA = FT != 0 && d == 8;
    extern __shared__ __align__(16) float lds[];
    const int F = FT ? FT : a.F;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4, g4 = 4 * g;
    const bool meta_q = a.flags & SATRANS_META_Q, meta_k = a.flags & SATRANS_META_K;
    const bool same_tab = a.tab_q == a.tab_k;
    const bool relu_out = a.flags & SATRANS_RELU_OUT, use_res = !(a.flags & SATRANS_NO_RES);

    // ---- carve LDS: bf16 images first (16-byte aligned pieces), then fp32 vectors and the q / k / v row buffers ---------
    // per layer: four projection images, the MetaNet images of one or two tables, six LayerNorm vectors - NL equal blocks
    const int per_b = 4 * D * KD + (same_tab ? 1 : 2) * (U * KD + D * KU);            // bf16 elements of a layer (multiples of 8)
    __bf16* const img0 = reinterpret_cast<__bf16*>(lds);
    float* const vec0 = reinterpret_cast<float*>(img0 + (size_t)NL * per_b);
    float* p = vec0 + NL * 6 * D;
    auto take = [&](int cnt) { float* r = p; p += (cnt + 3) & ~3; return r; };
    const int rows = ((Tsamp * F + 15) >> 4) << 4;
    float* sq = take(rows * LD);
    float* sk = take(rows * LD);
    float* sv = take(rows * LD);
    float* sx = take(rows * LD);      // the tile's rows between two layers: output of layer l = input (and residual) of layer l + 1

    for (int l = 0; l < NL; ++l) {
        const satrans_layer_desc& al = sa.d[l];
        __bf16* wq = img0 + (size_t)l * per_b;
        stage_bf16(al.w_query, wq, D, D, KD, true);
        stage_bf16(al.w_key, wq + D * KD, D, D, KD, true);
        stage_bf16(al.w_value, wq + 2 * D * KD, D, D, KD, true);
        stage_bf16(al.w_out, wq + 3 * D * KD, D, D, KD, false);
        float* vec = vec0 + l * 6 * D;
        for (int i = threadIdx.x; i < D; i += blockDim.x) {
            vec[4 * D + i] = al.ln_g[i]; vec[5 * D + i] = al.ln_b[i];
            if (meta_q) { vec[i] = al.lnq_g[i]; vec[D + i] = al.lnq_b[i]; }
            if (meta_k) { vec[2 * D + i] = al.lnk_g[i]; vec[3 * D + i] = al.lnk_b[i]; }
        }
    }
    const int il_d = n * KD + 8 * g, il_u = n * KU + 8 * g;      // this lane's fragment inside an image with rows of D / U inputs
    const float inv_sqrt_d = 1.0f / sqrtf((float)d);
    const WorkRange wr = work_range(a.seg, a.S, Tsamp, gridDim.x, blockIdx.x);

    int pre = 0;
    for (int scen = 0; scen < a.S && pre < wr.g1; ++scen) {
      const int nt_s = tiles_of(a.seg, scen, Tsamp);
      const int t0 = max(wr.g0, pre) - pre, t1 = min(wr.g1, pre + nt_s) - pre;
      pre += nt_s;
      if (t0 >= t1) continue;
      // ---- this scenario's generated MetaNet weights (the previous tile loop ended on a barrier) ------------------
      for (int l = 0; l < NL; ++l) {
          const satrans_layer_desc& al = sa.d[l];
          __bf16* w1q = img0 + (size_t)l * per_b + 4 * D * KD;
          __bf16* w2q = w1q + U * KD;
          if (meta_q) {
              const float* row = al.tab_q + (size_t)scen * al.tab_stride;
              stage_bf16(row, w1q, U, D, KD, true);               // W1 [D][U]: in-major, K = D
              stage_bf16(row + D * U, w2q, D, U, KU, true);       // W2 [U][D]: in-major, K = U
          }
          if (meta_k && (!same_tab || !meta_q)) {
              const float* row = al.tab_k + (size_t)scen * al.tab_stride;
              __bf16* w1k = same_tab ? w1q : w2q + D * KU;
              stage_bf16(row, w1k, U, D, KD, true);
              stage_bf16(row + D * U, w1k + U * KD, D, U, KU, true);
          }
      }
      __syncthreads();
      const int lo = a.seg[scen], hi = a.seg[scen + 1];
      for (int tile = t0; tile < t1; ++tile) {
        const int first = lo + tile * Tsamp;
        const int32_t* samp = a.order + first;
        const int nS = min(Tsamp, hi - first), ntok = nS * F, ntt = (ntok + 15) >> 4;

        for (int l = 0; l < NL; ++l) {
        // this layer's images and vectors
        const __bf16* wq = img0 + (size_t)l * per_b;
        const __bf16* wk = wq + D * KD; const __bf16* wv = wk + D * KD; const __bf16* wo = wv + D * KD;
        const __bf16* w1q = wo + D * KD; const __bf16* w2q = w1q + U * KD;
        const __bf16* w1k = same_tab ? w1q : w2q + D * KU; const __bf16* w2k = w1k + U * KD;
        const float* lnq_g = vec0 + l * 6 * D; const float* lnq_b = lnq_g + D; const float* lnk_g = lnq_b + D;
        const float* lnk_b = lnk_g + D; const float* ln_g = lnk_b + D; const float* ln_b = ln_g + D;
        const bool first_l = l == 0, last_l = l == NL - 1;
        // ---- phase 1: projections + MetaNet per 16-token tile, all in registers --------------------------
        for (int tt = wave; tt < ntt; tt += WAVES) {
            const int tok = 16 * tt + n;
            const bool valid = tok < ntok;
            const int ls = valid ? tok / F : 0, f = valid ? tok - ls * F : 0;
            const int b = samp[ls];
            float x[KT][4], q[KT][4], k[KT][4], v[KT][4];
            load_frag<KT>((first_l ? layer_x_row(a, b, f, F, D) : sx + (size_t)tok * LD) + g4, x);
            chain_bf16<KT, KT, KD>(wq + il_d, x, q);                                     // satrans.py:55-57
            chain_bf16<KT, KT, KD>(wk + il_d, x, k);
            chain_bf16<KT, KT, KD>(wv + il_d, x, v);
            auto metanet = [&](float (&z)[KT][4], const __bf16* w1, const __bf16* w2, const float* gam, const float* bet) {
                float h[UT][4], o[KT][4];                                                // submodules.py:77-103
                chain_bf16<KT, UT, KD>(w1 + il_d, z, h);
#pragma unroll
                for (int t = 0; t < UT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[t][r] = fmaxf(h[t][r], 0.f);
                chain_bf16<UT, KT, KU>(w2 + il_u, h, o);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[t][r] += o[t][r];
                float mean, rstd;
                layer_norm_frag<KT>(z, gam, bet, g4, mean, rstd);
            };
            if (meta_q) metanet(q, w1q, w2q, lnq_g, lnq_b);                              // satrans.py:60-66
            if (meta_k) metanet(k, w1k, w2k, lnk_g, lnk_b);                              // satrans.py:67-73
            store_frag<KT>(sq + (size_t)tok * LD + g4, q);
            store_frag<KT>(sk + (size_t)tok * LD + g4, k);
            store_frag<KT>(sv + (size_t)tok * LD + g4, v);
        }
        __syncthreads();

        // ---- phase 2: attention in fp32, one lane per (sample, head, query row)  (satrans.py:75-90) -------------------
        // two passes over the keys in chunks of four (maximum of the scaled scores, then exp2 / sum / PV with the scores
        // recomputed); padding keys of the last chunk read the last real row and are masked
        if constexpr (MFA) {
          constexpr int FP = (FT + 3) & ~3, NJB = FP / 4, SHW = 64 / FP;      // SHW whole (sample, head) groups per wave
          const int grp = lane / FP, m_i = lane - grp * FP, sub = lane & 3;   // (FP is a multiple of 4: m_i & 3 = lane & 3)
          const float sc_scale = inv_sqrt_d * kLog2e;
          const int ngrp = nS * H;
          for (int g0 = wave * SHW; g0 < ngrp; g0 += WAVES * SHW) {           // (wave-uniform: every lane runs the products)
            // spare lanes of the wave and groups beyond the tile repeat a real group: same operands, nothing stored
            const int sh = min(g0 + min(grp, SHW - 1), ngrp - 1);
            const bool own = grp < SHW && g0 + grp < ngrp && m_i < FT;
            const int ls = sh / H, h = sh - ls * H, iq = min(m_i, FT - 1);
            float* qrow = sq + (size_t)(ls * FT + iq) * LD + h * d;
            float qe[d];
            {
                const float4 q0_ = *reinterpret_cast<const float4*>(qrow), q1_ = *reinterpret_cast<const float4*>(qrow + 4);
                qe[0] = q0_.x; qe[1] = q0_.y; qe[2] = q0_.z; qe[3] = q0_.w; qe[4] = q1_.x; qe[5] = q1_.y; qe[6] = q1_.z; qe[7] = q1_.w;
            }
            float ka[NJB][d];
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) {      // (a padding key of the last block reads a row of the buffer: masked below)
                const float* kr = sk + (size_t)min(ls * FT + 4 * jb + sub, rows - 1) * LD + h * d;
                const float4 k0_ = *reinterpret_cast<const float4*>(kr), k1_ = *reinterpret_cast<const float4*>(kr + 4);
                ka[jb][0] = k0_.x; ka[jb][1] = k0_.y; ka[jb][2] = k0_.z; ka[jb][3] = k0_.w;
                ka[jb][4] = k1_.x; ka[jb][5] = k1_.y; ka[jb][6] = k1_.z; ka[jb][7] = k1_.w;
            }
            f32x2 vp[FT];
            {
                const float* vb = sv + (size_t)(ls * FT) * LD + h * d + 2 * sub;
#pragma unroll
                for (int j = 0; j < FT; ++j) vp[j] = *reinterpret_cast<const f32x2*>(vb + (size_t)j * LD);
            }
            f32x4 sc4[NJB];
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) sc4[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < d; ++e)
#pragma unroll
                for (int jb = 0; jb < NJB; ++jb)
                    sc4[jb] = __builtin_amdgcn_mfma_f32_4x4x1f32(ka[jb][e], qe[e], sc4[jb], 0, 0, 0);
            float ex[FT];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < FT; ++j) {
                ex[j] = sc4[j >> 2][j & 3] * sc_scale;
                mx = fmaxf(mx, ex[j]);
            }
            float sum = 0.f;
            f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < FT; ++j) {
                ex[j] = __builtin_amdgcn_exp2f(ex[j] - mx);
                sum += ex[j];
                oa = __builtin_amdgcn_mfma_f32_4x4x1f32(vp[j].x, ex[j], oa, 0, 0, 0);
                ob = __builtin_amdgcn_mfma_f32_4x4x1f32(vp[j].y, ex[j], ob, 0, 0, 0);
            }
            // the attention output takes the place of this lane's own q row slice (nobody else reads it)
            const float inv = 1.0f / sum;
            if (own) {
                *reinterpret_cast<float4*>(qrow) = make_float4(oa[0] * inv, ob[0] * inv, oa[1] * inv, ob[1] * inv);
                *reinterpret_cast<float4*>(qrow + 4) = make_float4(oa[2] * inv, ob[2] * inv, oa[3] * inv, ob[3] * inv);
            }
          }
        } else {
          const float sc_scale = inv_sqrt_d * kLog2e;
          for (int task = threadIdx.x; task < nS * H * F; task += 64 * WAVES) {
            const int ls = task / (H * F), rem = task - ls * H * F;
            const int h = rem / F, i = rem - h * F;
            float* qrow = sq + (size_t)(ls * F + i) * LD + h * d;
            const float* kbase = sk + (size_t)(ls * F) * LD + h * d;
            const float* vbase = sv + (size_t)(ls * F) * LD + h * d;
            f32x2 qi[d / 2];
            load_row<d>(qrow, qi);
            float mx = -INFINITY;
#pragma unroll 1
            for (int j0 = 0; j0 < F; j0 += 4) {
                f32x2 kr[4][d / 2];
#pragma unroll
                for (int u = 0; u < 4; ++u) load_row<d>(kbase + (size_t)min(j0 + u, F - 1) * LD, kr[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) mx = fmaxf(mx, dot_row<d>(qi, kr[u]) * sc_scale);
            }
            f32x2 oacc[d / 2];
#pragma unroll
            for (int e = 0; e < d / 2; ++e) oacc[e] = f32x2{0.f, 0.f};
            float sum = 0.f;
#pragma unroll 1
            for (int j0 = 0; j0 < F; j0 += 4) {
                f32x2 kr[4][d / 2], vr[4][d / 2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    load_row<d>(kbase + (size_t)min(j0 + u, F - 1) * LD, kr[u]);
                    load_row<d>(vbase + (size_t)min(j0 + u, F - 1) * LD, vr[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float ex = j0 + u < F ? __builtin_amdgcn_exp2f(dot_row<d>(qi, kr[u]) * sc_scale - mx) : 0.f;
                    sum += ex;
                    axpy_row<d>(ex, vr[u], oacc);
                }
            }
            // the attention output takes the place of this task's own q row (nobody else reads it)
            store_row<d>(qrow, oacc, 1.0f / sum);
          }
        }
        __syncthreads();

        // ---- phase 3: Out_linear, residual, LayerNorm per 16-token tile (satrans.py:91-99) ------------
        for (int tt = wave; tt < ntt; tt += WAVES) {
            const int tok = 16 * tt + n;
            const bool valid = tok < ntok;
            const int ls = valid ? tok / F : 0, f = valid ? tok - ls * F : 0;
            const int b = samp[ls];
            float o[KT][4], u[KT][4], x[KT][4];
            load_frag<KT>(sq + (size_t)tok * LD + g4, o);
            chain_bf16<KT, KT, KD>(wo + il_d, o, u);
            load_frag<KT>((first_l ? layer_x_row(a, b, f, F, D) : sx + (size_t)tok * LD) + g4, x);
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float val = u[t][r];
                    if (relu_out) val = fmaxf(val, 0.f);
                    if (use_res) val += x[t][r];
                    u[t][r] = val;
                }
            float mean, rstd;
            layer_norm_frag<KT>(u, ln_g, ln_b, g4, mean, rstd);
            if (last_l && !hd.prob) { if (valid) store_frag<KT>(y + ((size_t)b * F + f) * D + g4, u); }
            else store_frag<KT>(sx + (size_t)tok * LD + g4, u);
        }
        __syncthreads();
        if (last_l && hd.prob) {
            const int FD4 = F * (D / 4);
            const float4* w4 = reinterpret_cast<const float4*>(hd.w);
            for (int ls = wave; ls < nS; ls += WAVES) {
                const int b = samp[ls];
                float acc = 0.f;
                for (int i = lane; i < FD4; i += 64) {
                    const int f = i / (D / 4), col = (i - f * (D / 4)) * 4;
                    const float4 xv = *reinterpret_cast<const float4*>(sx + (size_t)(ls * F + f) * LD + col), wv = w4[i];
                    acc = fmaf(xv.x, wv.x, acc);
                    acc = fmaf(xv.y, wv.y, acc);
                    acc = fmaf(xv.z, wv.z, acc);
                    acc = fmaf(xv.w, wv.w, acc);
                }
                for (int j = lane; j < hd.n_dense; j += 64)
                    acc = fmaf(hd.dense[(size_t)b * hd.dense_stride + hd.dense_col[j & 1]], hd.w[F * D + j], acc);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
                const float z = acc + hd.bias[0];
                const float pr = 1.0f / (1.0f + expf(-z));
                if (lane == 0) {
                    hd.prob[b] = pr;
                    if (hd.logit) hd.logit[b] = z;
                }
            }
            __syncthreads();      // (the next tile's first layer writes these rows again)
        }
        }      // layers
      }
    }
}


static int64_t bf16_fwd_lds_bytes(int T, int F, int D, int U, bool same_tab) {
    const int64_t KD = D + 8, KU = U + 8, LD = D + 4;
    const int64_t rows = (((int64_t)T * F + 15) / 16) * 16;
    const int64_t bf = 4 * D * KD + (same_tab ? 1 : 2) * ((int64_t)U * KD + D * KU);
    return 2 * bf + 4 * (6 * D + 3 * rows * LD) + 256;
}

template <int D, int U, int H, int WAVES, int FT = 0>
static int launch_fwd_bf16(const satrans_layer_desc* d, float* y, hipStream_t stream) {
    const bool same_tab = d->tab_q == d->tab_k;
    int best = 0;
    double best_eff = 0.0;
    for (int t = 1; t <= 4 * WAVES; ++t) {
        if (bf16_fwd_lds_bytes(t, d->F, D, U, same_tab) > 156 * 1024) break;
        const int tok = t * d->F, ntt = (tok + 15) / 16;
        double eff = (double)tok / (16.0 * ntt) * (double)ntt / (double)(ceil_div(ntt, WAVES) * WAVES);
        if (FT != 0) {
            // matrix-pipe attention: whole rounds of WAVES * (64 / FP) (sample, head) groups; the attention is ~2/3 of the kernel
            const int per_round = WAVES * (64 / ((FT + 3) & ~3)), groups = t * H;
            const double eff_att = (double)groups / (double)(ceil_div(groups, per_round) * per_round);
            eff = 1.0 / (0.35 / eff + 0.65 / eff_att);
        }
        if (eff >= best_eff) { best_eff = eff; best = t; }
    }
    SATRANS_REQUIRE(best > 0, SATRANS_E_UNSUPPORTED, "layer_fwd(bf16): F=%d does not fit LDS", d->F);
    const size_t lds = (size_t)bf16_fwd_lds_bytes(best, d->F, D, U, same_tab);
    static size_t attr_set = 0;
    if (lds > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)layer_fwd_bf16_kernel<D, U, H, WAVES, FT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "layer_fwd(bf16): LDS attribute: %s", hipGetErrorString(e));
        attr_set = lds;
    }
    const int64_t tiles = ceil_div(d->B, best) + d->S;
    const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, (int64_t)cu_count()));
    layer_fwd_bf16_kernel<D, U, H, WAVES, FT><<<gx, 64 * WAVES, lds, stream>>>(*d, best, y);
    SATRANS_CHECK_LAUNCH("layer_fwd_bf16_kernel");
    return SATRANS_OK;
}

static int64_t bf16_stack_lds_bytes(int NL, int T, int F, int D, int U, bool same_tab) {
    const int64_t KD = D + 8, KU = U + 8, LD = D + 4;
    const int64_t rows = (((int64_t)T * F + 15) / 16) * 16;
    const int64_t bf = 4 * D * KD + (same_tab ? 1 : 2) * ((int64_t)U * KD + D * KU);
    return 2 * bf * NL + 4 * (6 * D * NL + 4 * rows * LD) + 256;
}

template <int D, int U, int H, int WAVES, int FT = 0>
static int launch_stack_bf16(const Bf16Stack& sa, float* y, const Bf16Head& hd, hipStream_t stream) {
    const satrans_layer_desc* d = &sa.d[0];
    const bool same_tab = d->tab_q == d->tab_k;
    int best = 0;
    double best_eff = 0.0;
    for (int t = 1; t <= 4 * WAVES; ++t) {
        // (the whole 160 KB of a CU: three AliCCP layers with a tile of nine samples - eleven 16-token tiles for twelve waves, exactly
        //  one round of 36 (sample, head) groups in the attention - come to 163,840 bytes with the alignment slack)
        if (bf16_stack_lds_bytes(sa.L, t, d->F, D, U, same_tab) > 160 * 1024) break;
        const int tok = t * d->F, ntt = (tok + 15) / 16;
        double eff = (double)tok / (16.0 * ntt) * (double)ntt / (double)(ceil_div(ntt, WAVES) * WAVES);
        if (FT != 0) {
            const int per_round = WAVES * (64 / ((FT + 3) & ~3)), groups = t * H;
            const double eff_att = (double)groups / (double)(ceil_div(groups, per_round) * per_round);
            eff = 1.0 / (0.5 / eff + 0.5 / eff_att);
        }
        if (eff >= best_eff) { best_eff = eff; best = t; }
    }
    SATRANS_REQUIRE(best > 0, SATRANS_E_UNSUPPORTED, "stack_fwd(bf16): %d layers of F=%d do not fit LDS", sa.L, d->F);
    const size_t lds = (size_t)bf16_stack_lds_bytes(sa.L, best, d->F, D, U, same_tab);
    static size_t attr_set = 0;
    if (lds > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)stack_fwd_bf16_kernel<D, U, H, WAVES, FT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "stack_fwd(bf16): LDS attribute: %s", hipGetErrorString(e));
        attr_set = lds;
    }
    const int64_t tiles = ceil_div(d->B, best) + d->S;
    const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, (int64_t)cu_count()));
    stack_fwd_bf16_kernel<D, U, H, WAVES, FT><<<gx, 64 * WAVES, lds, stream>>>(sa, best, y, hd);
    SATRANS_CHECK_LAUNCH("stack_fwd_bf16_kernel");
    return SATRANS_OK;
}

}  // namespace satrans

using namespace satrans;

// The stack of n layers (satrans.py:236-239) as one launch: built for D = 32 (the AliCCP / Alimama shapes), 1 <= n <= 4 layers of
// one shape, one set of flags and one scenario bucketing; layer 0 reads its rows as satrans_layer_fwd_bf16 does (fused gather
// included), the others read what the layer before them left in LDS (their x / x_rows are ignored).
extern "C" int satrans_stack_fwd_bf16_supported(int n, const satrans_layer_desc* const* descs) {
    if (!descs || n < 1 || n > 4) return 0;
    for (int l = 0; l < n; ++l) {
        const satrans_layer_desc* d = descs[l];
        if (!d || !satrans_layer_fwd_bf16_supported(d) || d->D != 32) return 0;
        const satrans_layer_desc* d0 = descs[0];
        if (d->B != d0->B || d->F != d0->F || d->H != d0->H || d->U != d0->U || d->S != d0->S || d->flags != d0->flags ||
            d->order != d0->order || d->seg != d0->seg || (d->tab_q == d->tab_k) != (d0->tab_q == d0->tab_k) ||
            d->tab_stride != d0->tab_stride)
            return 0;
    }
    const bool same_tab = descs[0]->tab_q == descs[0]->tab_k;
    return bf16_stack_lds_bytes(n, 1, descs[0]->F, 32, 64, same_tab) <= 160 * 1024;
}

static int stack_fwd_bf16_any(int n, const satrans_layer_desc* const* descs, float* y, const Bf16Head& hd, hipStream_t stream) {
    Bf16Stack sa;
    for (int l = 0; l < 4; ++l) sa.d[l] = *descs[l < n ? l : 0];
    sa.L = n;
    const satrans_layer_desc* d = descs[0];
    if (d->F == 19) return launch_stack_bf16<32, 64, 4, 12, 19>(sa, y, hd, stream);      // AliCCP
    if (d->F == 15) return launch_stack_bf16<32, 64, 4, 12, 15>(sa, y, hd, stream);      // Alimama
    return launch_stack_bf16<32, 64, 4, 12>(sa, y, hd, stream);
}

extern "C" int satrans_stack_fwd_bf16(int n, const satrans_layer_desc* const* descs, float* y, void* stream_) {
    SATRANS_REQUIRE(satrans_stack_fwd_bf16_supported(n, descs), SATRANS_E_UNSUPPORTED,
                    "stack_fwd(bf16): 1 - 4 evaluation layers of (D,U,H) = (32,64,4), one shape, flags and scenario bucketing");
    SATRANS_REQUIRE(y, SATRANS_E_BADARG, "stack_fwd(bf16): null output");
    Bf16Head hd = {};
    return stack_fwd_bf16_any(n, descs, y, hd, (hipStream_t)stream_);
}

// ... with the head: h->w [F*D + n_dense], h->bias, h->dense / dense_stride / h_dense_cols (HOST array, n_dense <= 2) as in
// satrans_layer_bwd_head; outputs h->prob [B] and h->logit [B] (optional).  The last layer's rows are not written anywhere.
extern "C" int satrans_stack_fwd_bf16_head(int n, const satrans_layer_desc* const* descs, const satrans_head_desc* h, void* stream_) {
    SATRANS_REQUIRE(satrans_stack_fwd_bf16_supported(n, descs), SATRANS_E_UNSUPPORTED,
                    "stack_fwd_head(bf16): 1 - 4 evaluation layers of (D,U,H) = (32,64,4), one shape, flags and scenario bucketing");
    SATRANS_REQUIRE(h && h->w && h->bias && h->prob && h->n_dense >= 0 && h->n_dense <= 2 && (h->n_dense == 0 || (h->dense && h->h_dense_cols)),
                    SATRANS_E_BADARG, "stack_fwd_head(bf16): bad head operands");
    Bf16Head hd = {};
    hd.w = h->w; hd.bias = h->bias; hd.dense = h->dense; hd.dense_stride = h->dense_stride; hd.n_dense = h->n_dense;
    for (int j = 0; j < h->n_dense; ++j) hd.dense_col[j] = h->h_dense_cols[j];
    hd.prob = h->prob; hd.logit = h->logit;
    return stack_fwd_bf16_any(n, descs, nullptr, hd, (hipStream_t)stream_);
}

extern "C" int satrans_layer_fwd_bf16_supported(const satrans_layer_desc* d) {
    if (!d || (d->flags & (SATRANS_GATE | SATRANS_BILINEAR | SATRANS_TRAIN))) return 0;
    const bool meta = d->flags & (SATRANS_META_Q | SATRANS_META_K);
    if (d->D == 32 && d->H == 4 && (!meta || d->U == 64)) return 1;
    if (d->D == 64 && d->H == 4 && (!meta || d->U == 128) && d->F <= 64) return 1;
    return 0;
}

extern "C" int satrans_layer_fwd_bf16(const satrans_layer_desc* d, float* y, void* stream_) {
    SATRANS_REQUIRE(satrans_layer_fwd_bf16_supported(d), SATRANS_E_UNSUPPORTED,
                    "layer_fwd(bf16): evaluation forward of (D,U,H) = (32,64,4) or (64,128,4) without gate / bilinear");
    SATRANS_REQUIRE(y, SATRANS_E_BADARG, "layer_fwd(bf16): null output");
    hipStream_t stream = (hipStream_t)stream_;
    if (d->D == 32 && d->F == 19) return launch_fwd_bf16<32, 64, 4, 12, 19>(d, y, stream);      // AliCCP
    if (d->D == 32 && d->F == 15) return launch_fwd_bf16<32, 64, 4, 12, 15>(d, y, stream);      // Alimama
    if (d->D == 32) return launch_fwd_bf16<32, 64, 4, 12>(d, y, stream);
    return launch_fwd_bf16<64, 128, 4, 8>(d, y, stream);
}
