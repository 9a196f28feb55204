// Meta_Transformer_Layer forward and backward, register-chained MFMA version for compile-time shapes.
//
// Reference: models/satrans.py:50-100 (layer), models/submodules.py:77-103 (MetaNet).  Same decomposition as
// layer_lds.hip (workgroup = one scenario, persistent over that scenario's tiles of T samples; backward recomputes
// the forward; weight gradients in per-workgroup slabs reduced in a fixed order), but built for speed:
//
//   * every weight matrix is staged into LDS ONCE per workgroup (the generated MetaNet weights of the workgroup's
//     scenario included) and stays there for all its tiles;
//   * the token-wise chain  x -> {q0,k0,v} -> relu(q0 W1) -> (.) W2 -> +q0 -> LayerNorm  runs per 16-token tile in
//     one wave entirely in registers: with tokens on the N side of v_mfma_f32_16x16x4_f32 the accumulator of one
//     product (lane: token = lane&15, features 16t + 4*(lane>>4) + r) IS the B operand of the next one, so no data
//     moves between the products.  f32-in/f32-accumulate MFMA is bit-for-bit an fmaf chain: fp32 parity is kept;
//   * LayerNorm statistics are a 8-register sum plus two cross-lane shuffles (lanes n, n+16, n+32, n+48 hold one
//     token);
//   * only q, k, v cross LDS (the feature x feature attention needs whole samples): one lane per
//     (sample, head, query row), keys/values read as LDS broadcasts, two passes (max, then exp / sum / PV), nothing
//     of size F x F is stored;
//   * shapes are template parameters (D, U, H): no runtime division in any inner loop.
#include <hip/hip_ext.h>

#include "layer_fused_common.h"

namespace satrans {

// satrans_kernel_timing: while armed, every launch of the fused layer kernels carries a pair of HIP events that the dispatch packet
// itself signals with its begin / end timestamps (hipExtLaunchKernelGGL) - the kernel's own duration on the stream it runs on,
// without the dispatch latency that events recorded around a launch include (5-15 us here, measured against rocprofv3).
struct KernelTimer {
    static constexpr int kPairs = 512;
    struct Pair { hipEvent_t start, stop; int kind; };
    Pair pairs[kPairs];
    int made = 0, used = 0;
    bool armed = false;
    Pair* next(int kind) {          // kind: 0 layer forward, 1 layer backward, 2 last layer + head in one launch
        if (!armed || used >= kPairs) return nullptr;
        if (used >= made) {
            if (hipEventCreate(&pairs[made].start) != hipSuccess || hipEventCreate(&pairs[made].stop) != hipSuccess) return nullptr;
            ++made;
        }
        pairs[used].kind = kind;
        return &pairs[used++];
    }
};
static KernelTimer g_ktimer;

// MOD: what modulates q / k - 0 the MetaNet (or nothing), 1 flag 'gate', 2 flag 'bilinear' (compile time: the main instantiation
// pays nothing for the other two).  Every product runs on v_mfma_f32_16x16x4_f32 (bit for bit an fmaf chain).
// SAVE: leave the attention's numerators / statistics / output (and the normalised MetaNet rows) in a.attn_save for the backward
// of this step
// Diagnostic build only (-DSATRANS_DIAG_FWDSAVE=mask): the training forward WITHOUT the stores of its hand-over's numerators (1),
// normalised MetaNet rows (2), 1 / sum + keep word + attention output (4) - what each part of the hand-over costs this kernel.
#ifdef SATRANS_DIAG_FWDSAVE
constexpr int kDiagFwdSave = SATRANS_DIAG_FWDSAVE;
#else
constexpr int kDiagFwdSave = 0;
#endif
// PROW: the hand-over's numerators as padded rows (see the store below)
template <int D, int U, int H, int WAVES = kFusedWaves, int MOD = 0, bool SAVE = false, bool PROW = false>
__global__ __launch_bounds__(64 * WAVES) void layer_fwd_fused_kernel(satrans_layer_desc a, int Tsamp,
                                                                      float* __restrict__ y, float* __restrict__ att) {
    constexpr int KT = D / 16, UT = U / 16, d = D / H, LD = D + 4, LU = U + 4;
    constexpr int SZ_DD = D * LD, SZ_W1 = D * LU, SZ_W2 = U * LD;      // floats of LDS per [K][LD]-style image
    extern __shared__ __align__(16) float lds[];
    const int F = a.F;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4, g4 = 4 * g;
    // flags 'gate' / 'bilinear' (satrans.py:61-64,68-71,79-81) replace the MetaNet: q, k *= 2 * vec[scenario] (vec = the generated
    // row, width D), or q_h = q_h M[scenario, h] (generated row = H matrices d x d).  They reuse the MetaNet's LDS: the doubled
    // gate vectors sit in the MetaNet LayerNorm slots, the bilinear maps as ONE block-diagonal D x D image in the W1 slot.
    constexpr bool gate = MOD == 1, bilin = MOD == 2;
    const bool meta_q = (a.flags & SATRANS_META_Q) && !bilin, meta_k = (a.flags & SATRANS_META_K) && !bilin;
    const bool mlp_q = meta_q && !gate, mlp_k = meta_k && !gate;          // the MetaNet proper
    const bool same_tab = a.tab_q == a.tab_k;

    // ---- carve LDS, stage the scenario-independent weights once --------------------------------------------
    float* p = lds;
    auto take = [&](int cnt) { float* r = p; p += (cnt + 3) & ~3; return r; };
    FwdImages<D, U> W;
    W.wq = take(SZ_DD); W.wk = take(SZ_DD); W.wv = take(SZ_DD); W.woT = take(SZ_DD);
    W.w1q = take(SZ_W1); W.w2q = take(SZ_W2);
    W.w1k = same_tab ? W.w1q : take(SZ_W1);
    W.w2k = same_tab ? W.w2q : take(SZ_W2);
    W.lnq_g = take(D); W.lnq_b = take(D); W.lnk_g = take(D); W.lnk_b = take(D); W.ln_g = take(D); W.ln_b = take(D);
    // (+ 3: the attention phase reads keys in chunks of four without clamping the last chunk - rows past a sample's F-th belong to
    //  the next sample or to the zero-initialised / stale tail of the buffer: finite values that meet a weight of exactly 0)
    const int rows = ((Tsamp * F + 3 + 15) >> 4) << 4;
    float* sq = take(rows * LD);
    float* sk = take(rows * LD);
    float* sv = take(rows * LD);
    float* sx = take(rows * LD);      // the tile's input rows: read from global memory ONCE (phase 1), re-read here for the residual

    {
        const ImageJob jobs[4] = {{a.w_query, W.wq, D, D, LD, false}, {a.w_key, W.wk, D, D, LD, false}, {a.w_value, W.wv, D, D, LD, false},
                                  {a.w_out, W.woT, D, D, LD, true}};      // woT[i][o] = Wo[o][i]  (nn.Linear: y = x @ Wo^T)
        stage_image_batch<4, (D * D / 4 + 64 * WAVES - 1) / (64 * WAVES), false>(jobs);
    }
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        W.ln_g[i] = a.ln_g[i]; W.ln_b[i] = a.ln_b[i];
        if (mlp_q) { W.lnq_g[i] = a.lnq_g[i]; W.lnq_b[i] = a.lnq_b[i]; }
        if (mlp_k) { W.lnk_g[i] = a.lnk_g[i]; W.lnk_b[i] = a.lnk_b[i]; }
    }
    for (int i = threadIdx.x; i < 3 * rows * LD; i += blockDim.x) sq[i] = 0.f;      // (q, k, v rows: finite from the start, see `rows`)

    const int wl = n;  // per-lane offset inside an image: row 4g, column n
    const float* wq_l = W.wq + g4 * LD + wl;
    const float* wk_l = W.wk + g4 * LD + wl;
    const float* wv_l = W.wv + g4 * LD + wl;
    const float* wo_l = W.woT + g4 * LD + wl;
    const float* w1q_l = W.w1q + g4 * LU + wl;
    const float* w2q_l = W.w2q + g4 * LD + wl;
    const float* w1k_l = W.w1k + g4 * LU + wl;
    const float* w2k_l = W.w2k + g4 * LD + wl;
    const FusedDrop dc = fused_drop(a);
    const float inv_sqrt_d = 1.0f / sqrtf((float)d);   // scores * (1/sqrt d): within 1 ulp of the reference's true division
    // Work split by SAMPLES, not by tiles: workgroup w owns the sorted sample positions [B w / G, B (w+1) / G) and walks its share
    // of a scenario in tiles of Tsamp samples plus a remainder.  The token phases are bound by the MFMA pipe that the waves of
    // a SIMD share, so an iteration costs what its busiest SIMD holds: 8192 samples in tiles of ten are 3.2 tiles per
    // workgroup, i.e. four full iterations for a fifth of the chip while the rest idles; 10 + 10 + 10 + 2 samples everywhere
    // is three full iterations and one with a single wave per SIMD.
    const int p0 = (int)((int64_t)a.B * blockIdx.x / gridDim.x), p1 = (int)((int64_t)a.B * (blockIdx.x + 1) / gridDim.x);

    STAMP_DECL          // (SATRANS_STAMPS builds: slots 9-12 = staging / phase 1 / phase 2 / phase 3 of this kernel)
    for (int scen = 0; scen < a.S; ++scen) {
      const int lo = max(a.seg[scen], p0), hi = min(a.seg[scen + 1], p1);
      if (lo >= hi) continue;
      // ---- this scenario's generated MetaNet weights (the previous tile loop ended on a barrier) ------------------
      if (mlp_q) {
          const float* row = a.tab_q + (size_t)scen * a.tab_stride;
          const ImageJob jobs[2] = {{row, W.w1q, D, U, LU, false}, {row + D * U, W.w2q, U, D, LD, false}};
          stage_image_batch<2, (D * U / 4 + 64 * WAVES - 1) / (64 * WAVES), false>(jobs);
      }
      if (mlp_k && (!same_tab || !mlp_q)) {
          const float* row = a.tab_k + (size_t)scen * a.tab_stride;
          const ImageJob jobs[2] = {{row, W.w1k, D, U, LU, false}, {row + D * U, W.w2k, U, D, LD, false}};
          stage_image_batch<2, (D * U / 4 + 64 * WAVES - 1) / (64 * WAVES), false>(jobs);
      }
      if (gate) {
          for (int i = threadIdx.x; i < D; i += blockDim.x) {
              if (meta_q) W.lnq_g[i] = 2.0f * a.tab_q[(size_t)scen * a.tab_stride + i];
              if (meta_k) W.lnk_g[i] = 2.0f * a.tab_k[(size_t)scen * a.tab_stride + i];
          }
      }
      if (bilin) {                                   // block-diagonal image: wb[h d + i][h d + j] = M[scenario, h][i][j]
          const float* row = a.tab_q + (size_t)scen * a.tab_stride;
          for (int e = threadIdx.x; e < D * D; e += blockDim.x) {
              const int r = e / D, c = e - r * D;
              W.w1q[r * LD + c] = (r / d == c / d) ? row[(r / d) * d * d + (r % d) * d + (c % d)] : 0.f;
          }
      }
      __syncthreads();
      // The input row of a wave's (first) token tile is fetched one tile ahead, behind the output block of the previous tile:
      // sample index -> row address -> row are two dependent global loads (HBM misses for the first layer, whose rows come
      // straight from the embedding arena), which three waves per SIMD do not hide at the top of phase 1.
      auto fetch_x = [&](int first_, int tt_, float (&dst)[KT][4]) {
          const int tok_ = 16 * tt_ + n, ntok_ = min(Tsamp, hi - first_) * F;
          const bool valid_ = tok_ < ntok_;
          const int ls_ = valid_ ? tok_ / F : 0, f_ = valid_ ? tok_ - ls_ * F : 0;
          load_frag<KT>(layer_x_row(a, a.order[first_ + ls_], f_, F, D) + g4, dst);
      };
      for (int first = lo; first < hi; first += Tsamp) {
        const int32_t* samp = a.order + first;
        const int nS = min(Tsamp, hi - first), ntok = nS * F, ntt = (ntok + 15) >> 4;

        STAMP(9);
        // ---- phase 1: projections + MetaNet per 16-token tile, all in registers --------------------------
        for (int tt = wave; tt < ntt; tt += WAVES) {
            const int tok = 16 * tt + n;
            const bool valid = tok < ntok;
            const int ls = valid ? tok / F : 0, f = valid ? tok - ls * F : 0;
            const int b = samp[ls];
            float x[KT][4], q[KT][4], k[KT][4], v[KT][4];
            fetch_x(first, tt, x);
            store_frag<KT>(sx + (size_t)tok * LD + g4, x);
            chain<KT, KT, LD>(wq_l, x, q);                                               // satrans.py:55-57
            chain<KT, KT, LD>(wk_l, x, k);
            chain<KT, KT, LD>(wv_l, x, v);
            float mean, rstd;
            if (gate) {                                                                    // satrans.py:61-62,68-69
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const float4 vq = *reinterpret_cast<const float4*>(W.lnq_g + 16 * t + g4);
                    const float4 vk = *reinterpret_cast<const float4*>(W.lnk_g + 16 * t + g4);
                    if (meta_q) { q[t][0] *= vq.x; q[t][1] *= vq.y; q[t][2] *= vq.z; q[t][3] *= vq.w; }
                    if (meta_k) { k[t][0] *= vk.x; k[t][1] *= vk.y; k[t][2] *= vk.z; k[t][3] *= vk.w; }
                }
            } else if (bilin) {                                                            // satrans.py:79-81
                float qb[KT][4];
                chain<KT, KT, LD>(W.w1q + g4 * LD + wl, q, qb);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) q[t][r] = qb[t][r];
            }
            // SAVE: the normalised MetaNet rows and their 1 / std go to the backward of this step as well (it then skips the two W2
            // products and the LayerNorm statistics of its recomputation): [sorted position][field][role][D] behind the attention
            // state, then 1 / std [sorted position][field][role]
            constexpr bool ZSAVE = SAVE && MOD == 0;
            float zsave[KT][4];
            float* save_z = nullptr;
            float* save_r = nullptr;
            if constexpr (ZSAVE) {
                const size_t pos = (size_t)(first + ls) * F + f;
                const size_t p_floats_ = (size_t)a.B * (PROW ? (size_t)H * F * ((F + 3) & ~3) : (size_t)F * H * F);
                float* z_all = a.attn_save + p_floats_ + (size_t)a.B * (2 * H * F + (size_t)F * D);
                save_z = z_all + pos * 2 * D + g4;
                save_r = z_all + (size_t)a.B * F * 2 * D + pos * 2;
            }
            if (mlp_q) {                                                                  // satrans.py:60-66
                float h[UT][4], o[KT][4];
                metanet_frag<D, U>(w1q_l, w2q_l, W.lnq_g, W.lnq_b, g4, dc, kSiteMetaQ,
                                   drop_sample_key(dc.key[kSiteMetaQ], (uint32_t)b), f, q, h, o, mean, rstd, ZSAVE ? zsave : nullptr);
                if (ZSAVE && !(kDiagFwdSave & 2) && valid) {
                    store_frag<KT>(save_z, zsave);
                    if (g == 0) save_r[0] = rstd;
                }
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) q[t][r] = o[t][r];
            }
            if (mlp_k) {                                                                  // satrans.py:67-73
                float h[UT][4], o[KT][4];
                metanet_frag<D, U>(w1k_l, w2k_l, W.lnk_g, W.lnk_b, g4, dc, kSiteMetaK,
                                   drop_sample_key(dc.key[kSiteMetaK], (uint32_t)b), f, k, h, o, mean, rstd, ZSAVE ? zsave : nullptr);
                if (ZSAVE && !(kDiagFwdSave & 2) && valid) {
                    store_frag<KT>(save_z + D, zsave);
                    if (g == 0) save_r[1] = rstd;
                }
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) k[t][r] = o[t][r];
            }
            float* qd = sq + (size_t)tok * LD + g4;
            float* kd = sk + (size_t)tok * LD + g4;
            float* vd = sv + (size_t)tok * LD + g4;
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                *reinterpret_cast<float4*>(qd + 16 * t) = make_float4(q[t][0], q[t][1], q[t][2], q[t][3]);
                *reinterpret_cast<float4*>(kd + 16 * t) = make_float4(k[t][0], k[t][1], k[t][2], k[t][3]);
                *reinterpret_cast<float4*>(vd + 16 * t) = make_float4(v[t][0], v[t][1], v[t][2], v[t][3]);
            }
        }
        __syncthreads();

        STAMP(10);
        // ---- phase 2: attention, one lane per (sample, head, query row)  (satrans.py:75-90) -------------------
        if (F <= 4 * kRowChunks) {
          // One pass over the keys: the score row stays in registers (chunks of four keys, guarded by the uniform F),
          // so k is read once; exp2 of pre-scaled scores; padding keys of the last chunk read the last real row and
          // carry a score of -inf, i.e. a weight of exactly 0.
          const float sc_scale = inv_sqrt_d * kLog2e;
          for (int task = threadIdx.x; task < nS * H * F; task += 64 * WAVES) {
            const int ls = task / (H * F), rem = task - ls * H * F;
            const int h = rem / F, i = rem - h * F;
            const int b = samp[ls];
            float* qrow = sq + (size_t)(ls * F + i) * LD + h * d;
            const float* kbase = sk + (size_t)(ls * F) * LD + h * d;
            const float* vbase = sv + (size_t)(ls * F) * LD + h * d;
            f32x2 qi[d / 2];
            load_row<d>(qrow, qi);
            float sc[4 * kRowChunks];
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < kRowChunks; ++c) {
                if (4 * c < F) {
                    f32x2 kr[4][d / 2];
#pragma unroll
                    for (int u = 0; u < 4; ++u) load_row<d>(kbase + (size_t)(4 * c + u) * LD, kr[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float s_ = 4 * c + u < F ? dot_row<d>(qi, kr[u]) * sc_scale : -INFINITY;
                        sc[4 * c + u] = s_;
                        mx = fmaxf(mx, s_);
                    }
                }
            }
            f32x2 oacc[d / 2];
#pragma unroll
            for (int e = 0; e < d / 2; ++e) oacc[e] = f32x2{0.f, 0.f};
            float sum = 0.f;
            const uint32_t skey = drop_sample_key(dc.key[kSiteAttn], (uint32_t)b);
            const uint32_t block0 = drop_attn_elem(h, F, i, 0) >> 2;
            // a.attn_save: what the backward of this step would recompute, per SORTED sample position p = first + ls (the layout
            // the backward tiles walk): numerators [p][j][H F], then 1 / sum [p][H F], keep word [p][H F], attention output
            // [p][F][D]; lanes of one sample are consecutive tasks, so every store instruction writes runs of up to H F floats
            // PROW (the backward will take the hand-over straight into registers, fused_save_rows): the numerators as one
            // padded ROW per (position, head, query row), [p][H][F][FP], FP = 4 ceil(F / 4): five 16-byte stores per lane instead
            // of nineteen 4-byte ones (the scattered stores were 6 of this kernel's 83 us), and 16-byte loads in the backward.
            const int HF = H * F, FPs = (F + 3) & ~3;
            const size_t p_floats = (size_t)a.B * (PROW ? (size_t)H * F * FPs : (size_t)F * HF);
            float* save_p = !SAVE ? nullptr
                            : PROW ? a.attn_save + (((size_t)(first + ls) * H + h) * F + i) * FPs
                                        : a.attn_save + (size_t)(first + ls) * F * HF + rem;
            uint32_t keepw = 0xFFFFFFFFu;
#pragma unroll
            for (int c = 0; c < kRowChunks; ++c) {
                if (4 * c < F) {
                    f32x2 vr[4][d / 2];
#pragma unroll
                    for (int u = 0; u < 4; ++u) load_row<d>(vbase + (size_t)(4 * c + u) * LD, vr[u]);
                    const uint32_t kb = dc.on ? drop_keep4(skey, block0 + (uint32_t)c, dc.thresh) : 0xFu;
                    keepw &= ~((~kb & 0xFu) << (4 * c));
                    float ex4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float ex = __builtin_amdgcn_exp2f(sc[4 * c + u] - mx);
                        ex4[u] = ex;
                        sum += ex;
                        if (SAVE && !(kDiagFwdSave & 1) && !PROW && 4 * c + u < F) save_p[(size_t)(4 * c + u) * HF] = ex;
                        float pe = ex;
                        if (dc.on) pe = (kb >> u) & 1u ? ex * dc.scale : 0.f;
                        sc[4 * c + u] = pe;
                        axpy_row<d>(pe, vr[u], oacc);
                    }
                    // (keys beyond F carry exp2(-inf) = 0: the row's padding)
                    if (SAVE && !(kDiagFwdSave & 1) && PROW)
                        *reinterpret_cast<float4*>(save_p + 4 * c) = make_float4(ex4[0], ex4[1], ex4[2], ex4[3]);
                }
            }
            if constexpr (SAVE && !(kDiagFwdSave & 4)) {
                float* inv_all = a.attn_save + p_floats;
                float* keep_all = inv_all + (size_t)a.B * HF;
                float* o_all = keep_all + (size_t)a.B * HF;
                const size_t t_ = (size_t)(first + ls) * HF + rem;
                inv_all[t_] = 1.0f / sum;
                keep_all[t_] = __uint_as_float(keepw);
                store_row<d>(o_all + ((size_t)(first + ls) * F + i) * D + h * d, oacc, 1.0f / sum);
            }
            if (att) {   // normalized_att_scores [H,B,F,F], after dropout (satrans.py:87); rarely requested
                float* arow = att + (((size_t)h * a.B + b) * F + i) * F;
#pragma unroll
                for (int c = 0; c < kRowChunks; ++c)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (4 * c + u < F) arow[4 * c + u] = sc[4 * c + u] / sum;
            }
            // the attention output takes the place of this task's own q row (nobody else reads it)
            store_row<d>(qrow, oacc, 1.0f / sum);
          }
        } else {
          for (int task = threadIdx.x; task < nS * H * F; task += 64 * WAVES) {
              const int ls = task / (H * F), rem = task - ls * H * F;
              const int h = rem / F, i = rem - h * F;
              const int b = samp[ls];
              float qi[d];
              float* qrow = sq + (size_t)(ls * F + i) * LD + h * d;
  #pragma unroll
              for (int e = 0; e < d; e += 4) {
                  const float4 t4 = *reinterpret_cast<const float4*>(qrow + e);
                  qi[e] = t4.x; qi[e + 1] = t4.y; qi[e + 2] = t4.z; qi[e + 3] = t4.w;
              }
              const float* kbase = sk + (size_t)(ls * F) * LD + h * d;
              const float* vbase = sv + (size_t)(ls * F) * LD + h * d;
              float mx = -INFINITY;
  #pragma unroll 4
              for (int j = 0; j < F; ++j) {
                  float s = 0.f;
  #pragma unroll
                  for (int e = 0; e < d; e += 4) {
                      const float4 k4 = *reinterpret_cast<const float4*>(kbase + (size_t)j * LD + e);
                      s = fmaf(qi[e], k4.x, s); s = fmaf(qi[e + 1], k4.y, s);
                      s = fmaf(qi[e + 2], k4.z, s); s = fmaf(qi[e + 3], k4.w, s);
                  }
                  mx = fmaxf(mx, s * inv_sqrt_d);
              }
              float oacc[d];
  #pragma unroll
              for (int e = 0; e < d; ++e) oacc[e] = 0.f;
              float sum = 0.f;
              const uint32_t skey = drop_sample_key(dc.key[kSiteAttn], (uint32_t)b);
  #pragma unroll 4
              for (int j = 0; j < F; ++j) {
                  float s = 0.f;
  #pragma unroll
                  for (int e = 0; e < d; e += 4) {
                      const float4 k4 = *reinterpret_cast<const float4*>(kbase + (size_t)j * LD + e);
                      s = fmaf(qi[e], k4.x, s); s = fmaf(qi[e + 1], k4.y, s);
                      s = fmaf(qi[e + 2], k4.z, s); s = fmaf(qi[e + 3], k4.w, s);
                  }
                  const float ex = __expf(s * inv_sqrt_d - mx);
                  sum += ex;
                  float pe = ex;
                  if (dc.on) pe = drop_keep(skey, drop_attn_elem(h, F, i, j), dc.thresh) ? ex * dc.scale : 0.f;
  #pragma unroll
                  for (int e = 0; e < d; e += 4) {
                      const float4 v4 = *reinterpret_cast<const float4*>(vbase + (size_t)j * LD + e);
                      oacc[e] = fmaf(pe, v4.x, oacc[e]); oacc[e + 1] = fmaf(pe, v4.y, oacc[e + 1]);
                      oacc[e + 2] = fmaf(pe, v4.z, oacc[e + 2]); oacc[e + 3] = fmaf(pe, v4.w, oacc[e + 3]);
                  }
              }
              const float inv = 1.0f / sum;
              if (att) {   // normalized_att_scores [H,B,F,F], after dropout (satrans.py:87); rarely requested
                  float* arow = att + (((size_t)h * a.B + b) * F + i) * F;
      #pragma unroll 4
              for (int j = 0; j < F; ++j) {
                      float s = 0.f;
  #pragma unroll
                      for (int e = 0; e < d; e += 4) {
                          const float4 k4 = *reinterpret_cast<const float4*>(kbase + (size_t)j * LD + e);
                          s = fmaf(qi[e], k4.x, s); s = fmaf(qi[e + 1], k4.y, s);
                          s = fmaf(qi[e + 2], k4.z, s); s = fmaf(qi[e + 3], k4.w, s);
                      }
                      float pj = __expf(s * inv_sqrt_d - mx) / sum;
                      if (dc.on) pj = drop_keep(skey, drop_attn_elem(h, F, i, j), dc.thresh) ? pj * dc.scale : 0.f;
                      arow[j] = pj;
                  }
              }
              // the attention output takes the place of this task's own q row (nobody else reads it)
  #pragma unroll
              for (int e = 0; e < d; e += 4)
                  *reinterpret_cast<float4*>(qrow + e) =
                      make_float4(oacc[e] * inv, oacc[e + 1] * inv, oacc[e + 2] * inv, oacc[e + 3] * inv);
          }
        }
        __syncthreads();

        STAMP(11);
        // ---- phase 3: Out_linear, dropout, residual, LayerNorm per 16-token tile (satrans.py:91-99) ------------
        for (int tt = wave; tt < ntt; tt += WAVES) {
            const int tok = 16 * tt + n;
            const bool valid = tok < ntok;
            const int ls = valid ? tok / F : 0, f = valid ? tok - ls * F : 0;
            const int b = samp[ls];
            float o[KT][4], u[KT][4];
            const float* orow = sq + (size_t)tok * LD + g4;
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const float4 ov = *reinterpret_cast<const float4*>(orow + 16 * t);
                o[t][0] = ov.x; o[t][1] = ov.y; o[t][2] = ov.z; o[t][3] = ov.w;
            }
            chain<KT, KT, LD>(wo_l, o, u);
            const float* xrow = sx + (size_t)tok * LD + g4;
            const uint32_t skey = drop_sample_key(dc.key[kSiteOut], (uint32_t)b);
            const uint32_t kb = dc.on ? token_keep_bits<KT>(skey, f, D, g4, dc.thresh) : 0xFFFFFFFFu;
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const float4 xv = *reinterpret_cast<const float4*>(xrow + 16 * t);
                const float xr[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float val = u[t][r];
                    if (a.flags & SATRANS_RELU_OUT) val = fmaxf(val, 0.f);
                    if (dc.on) val = (kb >> (4 * t + r)) & 1u ? val * dc.scale : 0.f;
                    if (!(a.flags & SATRANS_NO_RES)) val += xr[r];
                    u[t][r] = val;
                }
            }
            float mean, rstd;
            layer_norm_frag<KT>(u, W.ln_g, W.ln_b, g4, mean, rstd);
            if (valid) {
                float* yrow = y + ((size_t)b * F + f) * D + g4;
#pragma unroll
                for (int t = 0; t < KT; ++t)
                    *reinterpret_cast<float4*>(yrow + 16 * t) = make_float4(u[t][0], u[t][1], u[t][2], u[t][3]);
            }
        }
        __syncthreads();
        STAMP(12);
      }
    }
}

// -------------------------------------------------------------------------------------------------------------------
// Backward.  One workgroup tile = T samples with T*F <= 64 tokens, i.e. at most ONE 16-token tile per wave, so all
// token-wise state of a tile (x, q0, k0, MetaNet hidden, normalised rows, statistics, dr) stays in the owning wave's
// registers across the attention barriers: nothing is recomputed twice, nothing is stashed.  Phases per tile:
//   A  (wave/tile)  forward chain, q k v -> LDS
//   B  (lane/task)  attention forward -> o, softmax statistics (max, 1/sum) per (sample, head, row)
//   C  (wave/tile)  Out_linear + residual + LayerNorm forward and backward, dWo, go = du Wo (replaces o)
//   D  (lane/task)  softmax backward by rows: dS_ij (cached), dq_i
//   E  (lane/task)  by columns from the cached dS and P: dk_j, dv_j (replace k_j, v_j)
//   F  (wave/tile)  MetaNet backward for Q and K, projection backward, all weight gradients, dx
// Weight gradients are MFMA products contracted over the tile's 16 tokens; their operands are written row-wise
// into the wave's OWN rows of the q / o / dq buffers (dead by then), so phase F needs no workgroup barrier.  The
// accumulators live in registers for the whole persistent loop and are combined over the four waves at the end.
// -------------------------------------------------------------------------------------------------------------------
// SAME: Q and K roles share one generated-weight table (no 'pos' flag).  TR: transposed copies of every weight image live in LDS
// too (conflict-free reads for the backward products); without them those products read the forward images by rows (chain_t).
// The key / query loops of the attention phases in chunks of four: with the field count as a template constant (FT) the chunks are
// unrolled - constant addresses, no bounds arithmetic, no branches - with a scheduling barrier after each, so that the loads of
// later chunks are not hoisted over earlier ones (that costs the registers the token state needs: 77 spilled VGPRs); FT = 0
// keeps the rolled loop over a.F.
// (measured per loop with -Rpass-analysis: unrolling the exp / P V loop of phase B is what spills - 56 VGPRs on its own - so that
// one stays rolled; the other four unroll spill-free)
// (dP of phase D in registers between its two passes: 60 spilled VGPRs, +10 % - measured, not built)
#define ATTN_CHUNKS(VAR, BODY, UNROLL)                                                                          \
    if constexpr (FT != 0 && (UNROLL)) {                                                                                  \
        _Pragma("unroll") for (int VAR = 0; VAR < FT; VAR += 4) { BODY(VAR); __builtin_amdgcn_sched_barrier(0); } \
    } else {                                                                                                    \
        _Pragma("unroll 1") for (int VAR = 0; VAR < F; VAR += 4) BODY(VAR);                                      \
    }

// The same loop as a two-stage software pipeline (field count a constant): the LDS reads of chunk c + 1 are issued before chunk c
// is computed - at one wave per SIMD nothing else hides an LDS round trip (five chunks, two passes: ten of them per phase).
#define ATTN_PIPE(BUFT, LOAD, COMP)                                                            \
    {                                                                                          \
        BUFT buf_a, buf_b;                                                                     \
        LOAD(0, buf_a);                                                                        \
        _Pragma("unroll") for (int c0_ = 0; c0_ < FT; c0_ += 8) {                              \
            if (c0_ + 4 < FT) LOAD(c0_ + 4, buf_b);                                            \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            COMP(c0_, buf_a);                                                                  \
            if (c0_ + 4 < FT) {                                                                \
                if (c0_ + 8 < FT) LOAD(c0_ + 8, buf_a);                                        \
                __builtin_amdgcn_sched_barrier(0);                                             \
                COMP(c0_ + 4, buf_b);                                                          \
            }                                                                                  \
        }                                                                                      \
    }

// FT: the field count as a constant (0 = a.F).  MOD: 0 MetaNet (or nothing), 1 flag 'gate', 2 flag 'bilinear'.
// token-contraction products (weight gradients) with conflict-free operand reads
#define WGRAD wgrad_r4
// SAVE: a.attn_save holds what the forward of this step left of the attention and of the MetaNet LayerNorms
// HEADF: the LAST layer of a training step with the head fused in (satrans_layer_bwd_head): the output rows y this kernel
// recomputes anyway ARE the layer's forward, a tile holds whole samples, and flatten + Linear + sigmoid + loss + their backward
// (satrans.py:244-255, meta_basemodel.py:317) are sample-local - so the step needs neither a forward launch for this layer nor
// the head launches nor the [B,F,D] round trips of y and dy: logit = sum of the sample's token dots (one barrier in the middle
// of phase C), dy = dlogit * w_head in registers.  `dy` is not read; `hd` carries the head's operands.
struct FusedHeadArgs {
    const float* w;            // [F D + n_dense]
    const float* bias;         // [1]
    const float* labels;       // [B]
    const float* dense;        // float matrix holding the dense columns; never null (n_dense == 0: any readable address)
    int64_t dense_stride;      // (n_dense == 0: 0)
    int32_t dense_col[2];      // its columns in feature order, by value (no dependent index load in the kernel)
    int32_t n_dense, loss_kind;
    float *prob, *logit;       // [B]
    float* partial;            // [G][F D + n_dense + 2]: per-workgroup g_w | g_b | loss (rows of head_reduce_kernel)
};
#ifdef SATRANS_ATTN_PIPE
constexpr bool kAttnPipe = true;
#else
constexpr bool kAttnPipe = false;
#endif
// Attention backward on the matrix pipe (round 6): v_mfma_f32_4x4x1_16b_f32 is sixteen INDEPENDENT 4 x 4 rank-1 updates per
// instruction - block b multiplies the A values of lanes 4b..4b+3 with the B values of the same four lanes - at the rate of the
// other f32 shapes (8.8 cycles per instruction, one accumulator or several: tools/microbench/mfma_4x4.hip).  With one lane per
// (sample, head, query row), rows padded to a multiple of 4 so that a block = four consecutive rows of one (sample, head):
//   B = the lane's OWN operand (its go row, its dS row), A = shared rows addressed by lane & 3 -> register r of the lane is
//   element (key 4 jb + r) of ITS row of dP^T, or feature pair r of ITS dq row: the F = 19, d = 8 products run without
//   padding to a 16 x 16 tile (which wastes 2.8 x on the F side and 2 x on the d side) and the softmax backward between them
//   stays lane-local exactly as in the wavefront arm.  -DSATRANS_ATTN_LANE builds that arm instead (the ablation).
#ifdef SATRANS_ATTN_LANE
constexpr bool kAttnMfma = false;
#else
constexpr bool kAttnMfma = true;
#endif
// Diagnostic build only (-DSATRANS_DIAG_SKIP=mask): the backward kernel WITHOUT its phase D (1) / E (2) / B (4) bodies - wrong
// results, the same launch otherwise: the kernel's duration with a phase removed is that phase's cost without the stamps'
// own perturbation (tools/experiments/r06_phase_cost.sh).  Never compiled into the shipped library.
#ifdef SATRANS_DIAG_SKIP
constexpr int kDiagSkip = SATRANS_DIAG_SKIP;
#else
constexpr int kDiagSkip = 0;
#endif
__device__ __forceinline__ f32x4 mfma_b16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}
constexpr int kHeadDenseMax = 2;      // dense columns the fused head carries (Alimama: 1); more -> the separate head launches
template <int D, int U, int H, bool SAME, bool TR, int FT = 0, int MOD = 0, bool SAVE = false, bool HEADF = false>
__global__ __launch_bounds__(kFusedBlock, 1) void layer_bwd_fused_kernel(satrans_layer_desc a, int Tsamp,
                                                                         const float* __restrict__ dy,
                                                                         float* __restrict__ dx,
                                                                         float* __restrict__ slabs, FusedHeadArgs hd) {
    constexpr int KT = D / 16, UT = U / 16, d = D / H, LD = D + 4, LU = U + 4;
    constexpr int HB = (UT + KT - 1) / KT;          // row buffers needed to hold one U-wide operand (<= 2)
    constexpr int NB = (UT < KT) ? UT : KT;          // 16-feature tiles of such an operand held by one row buffer
    static_assert(HB <= 2 && UT == HB * NB, "MetaNet hidden width must be D/.. or 2*D for the fused backward");
    static_assert(KT <= 2, "the cached dropout keep flags hold 8 bits per site");
    static_assert(64 * H <= kFusedBlock, "one attention task per thread: a tile holds at most 64 tokens x H heads");
    constexpr int SZ_DD = D * LD, SZ_W1 = D * LU, SZ_W2 = U * LD;
    // the matrix-pipe arm of phases D / E: field count a constant, rows padded to FP, one lane per padded (sample, head, row)
    // Lanes: a wave holds SHW = 64 / FP whole (sample, head) groups of FP lanes (F = 19: 3 groups of 20, four lanes spare; F = 15:
    // 4 of 16), so that no group straddles two waves - everything phases D and E exchange about a (sample, head) stays inside one
    // wave, and E follows D behind a wave-local wait instead of a workgroup barrier.
    constexpr int FP = (FT + 3) & ~3, NJB = FP / 4, SHW = FT ? 64 / (FP ? FP : 1) : 1;
    constexpr bool MFA = kAttnMfma && FT != 0 && d == 8 && (64 / (FT ? FT : 64)) * H <= kFusedWaves * SHW;
    extern __shared__ __align__(16) float lds[];
    const int F = FT ? FT : a.F;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4, g4 = 4 * g;
    // flags 'gate' / 'bilinear' replace the MetaNet (see the forward kernel): doubled gate vectors in the MetaNet LayerNorm slots,
    // the bilinear maps as one block-diagonal D x D image in the W1 slot; their gradients accumulate in the (then unused) MetaNet
    // LayerNorm / W1 accumulators and leave through the scenario records
    constexpr bool gate = MOD == 1, bilin = MOD == 2;
    const bool meta_q = (a.flags & SATRANS_META_Q) && !bilin, meta_k = (a.flags & SATRANS_META_K) && !bilin;
    const bool mlp_q = meta_q && !gate, mlp_k = meta_k && !gate;          // the MetaNet proper
    constexpr bool same_tab = SAME;
    // SAME = one generated-weight table AND one MetaNet LayerNorm for the Q and K roles (no 'pos' flag: satrans.py:46 - the
    // launcher checks both): the LayerNorm gradients of both roles then share one set of accumulators, as the W1 / W2 ones do
    constexpr bool SHLN = SAME && MOD == 0;
    const bool relu_out = a.flags & SATRANS_RELU_OUT, use_res = !(a.flags & SATRANS_NO_RES);

    // ---- LDS: forward images, transposed images, LN vectors, 5 row buffers, softmax cache ------------------------
    float* p = lds;
    auto take = [&](int cnt) { float* r = p; p += (cnt + 3) & ~3; return r; };
    float* wq = take(SZ_DD); float* wk = take(SZ_DD); float* wv = take(SZ_DD); float* woT = take(SZ_DD);
    float* w1q = take(SZ_W1); float* w2q = take(SZ_W2);
    float* w1k = same_tab ? w1q : take(SZ_W1);
    float* w2k = same_tab ? w2q : take(SZ_W2);
    // transposed copies (TR) or, without them, the forward images again (read by rows: chain_t)
    constexpr bool TWO = TR;
    float* wqT = TWO ? take(SZ_DD) : wq; float* wkT = TWO ? take(SZ_DD) : wk; float* wvT = TWO ? take(SZ_DD) : wv;
    float* wo = TWO ? take(SZ_DD) : woT;
    // w1T[u][i] = W1[i][u];  w2T[o][u] = W2[u][o]
    float* w1qT = TR ? take(U * LD) : w1q;
    float* w2qT = TR ? take(D * LU) : w2q;
    float* w1kT = TWO ? (same_tab ? w1qT : take(U * LD)) : w1k;
    float* w2kT = TWO ? (same_tab ? w2qT : take(D * LU)) : w2k;
    float* lnq_g = take(D); float* lnk_g = take(D); float* ln_g = take(D);
    float* lnq_b = take(D); float* lnk_b = take(D); float* ln_b = take(D);
    constexpr int ROWS = 64;
    float* sq = take(ROWS * LD);     // q      | phase F scratch
    float* sk = take(ROWS * LD);     // k      -> dk
    float* sv = take(ROWS * LD);     // v      -> dv
    float* so = take(ROWS * LD);     // o      -> go | phase F scratch
    float* sg = take(ROWS * LD);     // du (phase C scratch) -> dq | phase F scratch
    const int ntask_max = Tsamp * H * F;
    float* st_inv = take(ntask_max);                  // 1 / sum_j exp(s_ij - max_i)
    uint32_t* st_keep = (uint32_t*)take(ntask_max);   // bit j: attention-dropout keep flag of (i, j)   (F <= 32)
    float* sP = take(ntask_max * F);                  // exp(s_ij - max_i), the un-normalised softmax numerators
    // dP_ij, then dS_ij (phases D, E); SAVE: first the staging area of the tile's dy rows and saved attention outputs (2 T F D)
    // (matrix-pipe arm: dS transposed and padded, [sample, head][key][query row padded to FP] - phase D writes one float per key,
    //  phase E reads its key's row 16 bytes at a time)
    float* sDS = take(max(SAVE ? max(ntask_max * F, 2 * Tsamp * F * D) : ntask_max * F, MFA ? Tsamp * H * FP * FP : 0));
    // HEADF: head weights [F][D], the tile's token dots, per-sample dense terms / loss / dlogit, per-lane head-gradient sums
    float* s_wh = HEADF ? take(F * D + kHeadDenseMax) : nullptr;
    float* s_dot = HEADF ? take(64) : nullptr;
    float* s_dn = HEADF ? take(64) : nullptr;                            // dense-feature term of the logit per sample slot of the tile
    float* s_fin = HEADF ? take(2 * 64) : nullptr;                       // kernel end: loss / dlogit sums per sample slot
    float* s_hw = HEADF ? take(kFusedBlock * 4 * KT) : nullptr;          // 4 KT floats per lane: dW_head of the lane's (field, features)
    float* s_hwd = HEADF ? take(64 * kHeadDenseMax) : nullptr;           // dW of the dense columns per sample slot
    static_assert(!HEADF || !SAVE, "fused head: the recomputing backward (the last layer has no forward launch to save anything)");

    const WorkRange wr = work_range(a.seg, a.S, Tsamp, gridDim.x, blockIdx.x);
    const bool idle = wr.g0 >= wr.g1;      // no tile for this workgroup: only its zero slab is due
    if (!idle) {
        // (without transposed copies the images are swizzled: by-columns AND by-rows reads conflict-free, layer_fused_common.h)
        {
            const ImageJob jobs[4] = {{a.w_query, wq, D, D, LD, false}, {a.w_key, wk, D, D, LD, false}, {a.w_value, wv, D, D, LD, false},
                                      {a.w_out, woT, D, D, LD, true}};
            stage_image_batch<4, (D * D / 4 + kFusedBlock - 1) / kFusedBlock, !TR>(jobs);
        }
        if constexpr (TR) {
            stage_image(a.w_query, wqT, D, D, LD, true);
            stage_image(a.w_key, wkT, D, D, LD, true);
            stage_image(a.w_value, wvT, D, D, LD, true);
            stage_image(a.w_out, wo, D, D, LD, false);
        }
        for (int i = threadIdx.x; i < D; i += blockDim.x) {
            ln_g[i] = a.ln_g[i]; ln_b[i] = a.ln_b[i];
            if (mlp_q) { lnq_g[i] = a.lnq_g[i]; lnq_b[i] = a.lnq_b[i]; }
            if (mlp_k) { lnk_g[i] = a.lnk_g[i]; lnk_b[i] = a.lnk_b[i]; }
        }
        // rows of padding tokens are multiplied by exact zeros in the token-contraction products: they must hold
        // finite numbers from the start (0 * NaN would poison an accumulator)
        for (int i = threadIdx.x; i < 5 * ROWS * LD; i += blockDim.x) sq[i] = 0.f;
        if constexpr (HEADF) {
            for (int i = threadIdx.x; i < F * D + hd.n_dense; i += blockDim.x) s_wh[i] = hd.w[i];
            for (int i = threadIdx.x; i < kFusedBlock * 4 * KT; i += blockDim.x) s_hw[i] = 0.f;
            for (int i = threadIdx.x; i < 64 * kHeadDenseMax; i += blockDim.x) s_hwd[i] = 0.f;
            for (int i = threadIdx.x; i < 64; i += blockDim.x) s_dn[i] = 0.f;
        }
    }
    __syncthreads();

    // per-lane offsets into an image: by columns (chain: row 4g + r, column n) and by rows (chain_t: row n, column 4g); the
    // swizzle of the non-TR images flips one address bit per lane group / per row
    const int fl_g = (!TR && (g == 1 || g == 2)) ? 4 : 0, fl_n = (!TR && n >= 4 && n < 12) ? 1 : 0;
    const int lo_d = g4 * LD + (n ^ fl_g), lo_u = g4 * LU + (n ^ fl_g);
    const int lt_d = n * LD + 4 * (g ^ fl_n), lt_u = n * LU + 4 * (g ^ fl_n);
    // (sample, head, row) of this thread's attention task - the same in every tile and every attention phase
    const int t0_ls = (int)threadIdx.x / (H * F), t0_rem = (int)threadIdx.x - t0_ls * H * F;
    const int t0_h = t0_rem / F, t0_i = t0_rem - t0_h * F;
    // matrix-pipe arm: (sample, head, padded row) of this lane; lanes beyond the tile's samples address the last sample
    // (the spare lanes of a wave repeat the first rows of its last group: same operands, same results, same addresses)
    constexpr int FPd = MFA ? FP : 1;
    const int m_grp = lane / FPd, m_sh = wave * SHW + min(m_grp, SHW - 1);
    const int m_ls_raw = m_sh / H, m_h = m_sh - m_ls_raw * H, m_i = lane - m_grp * FPd, m_sub = lane & 3;      // (FP is a multiple of 4: m_i & 3 = lane & 3)
    auto ld_pair = [](const float* p_) -> f32x2 { return *reinterpret_cast<const f32x2*>(p_); };      // (a feature pair of a row)
    // (round 6, measured on one box: keeping the compiler from pairing these reads into ds_read2_b64 - half the LDS rate per
    //  byte - or pinning the reads ahead of the products with scheduling barriers moves the kernel by < 1 %: left to the compiler)
    const FusedDrop dc = fused_drop(a);
    const float inv_sqrt_d = 1.0f / sqrtf((float)d);
    // a.attn_save: the forward of this step left the attention's softmax numerators [p][j][H F], 1 / sum [p][H F], keep words
    // [p][H F] and outputs [p][F][D] per SORTED sample position p (layer_fwd_fused_kernel): phase B is then a copy - straight
    // into LDS, issued at the top of the tile and complete by the end of phase A - instead of a recomputation
    const int HF = H * F;
    constexpr bool has_save = SAVE;
    // REGH: the saved attention goes straight from HBM into the registers of the lanes that use it - the token lane's dy row and
    // saved attention output (issued at the top of the tile, used in phase C), the attention lane's softmax numerators, 1 / sum and
    // keep word (issued at the start of phase C, used in phase D) - instead of through LDS by eleven LDS-DMA pieces per wave and
    // tile: a piece costs 100-185 cycles to issue, a plain 16-byte load a few, and the LDS reads of the staged copies go too.
    // (The matrix-pipe arm only: its attention lanes are known per wave.)
#ifdef SATRANS_HANDOVER_LDS
    constexpr bool REGH = false;
#else
    constexpr bool REGH = SAVE && MFA;
#endif
    constexpr bool has_zsave = SAVE && MOD == 0;       // ... and the normalised MetaNet rows with their 1 / std
    // the MetaNet's hidden rows relu(z0 W1) are computed where phase F needs them: always with the saved rows (phase A has no use
    // for them then), and - recomputed, 2 x 32 more MFMAs per tile - in the one instantiation that otherwise spills 64 registers
    // (separate Q / K tables with the head fused in: 192 accumulator registers)
    constexpr bool rehidden = has_zsave || (HEADF && !SAME);
    // (written so that the instantiations without the register hand-over keep the expression - and the code - they had:
    //  tools/experiments/README.md, round 6, "a codegen-fragile instantiation")
    const float* save_inv = REGH ? a.attn_save + (size_t)a.B * H * F * FP : a.attn_save + (size_t)a.B * F * HF;
    const float* save_keep = save_inv + (size_t)a.B * HF;
    const float* save_o = save_keep + (size_t)a.B * HF;
    const float* save_z = save_o + (size_t)a.B * F * D;
    const float* save_r = save_z + (size_t)a.B * F * 2 * D;
    using lds_ptr = __attribute__((address_space(3))) void*;

    // ---- register accumulators of the weight gradients (whole kernel) ----------------------------------------------
    f32x4 acc_wq[KT][KT], acc_wk[KT][KT], acc_wv[KT][KT], acc_wo[KT][KT];
    f32x4 acc_w1q[KT][UT], acc_w2q[UT][KT], acc_w1k[KT][UT], acc_w2k[UT][KT];
    float agq[KT][4], abq[KT][4], agk[KT][4], abk[KT][4], agl[KT][4], abl[KT][4];
    float head_loss = 0.f, head_gb = 0.f;      // HEADF: loss / dlogit sums of the samples whose (field 0, group 0) lane this is
    float head_bias = 0.f;
    if constexpr (HEADF) head_bias = hd.bias[0];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KT; ++i) {
#pragma unroll
        for (int j = 0; j < KT; ++j) { acc_wq[i][j] = zero4; acc_wk[i][j] = zero4; acc_wv[i][j] = zero4; acc_wo[i][j] = zero4; }
#pragma unroll
        for (int j = 0; j < UT; ++j) { acc_w1q[i][j] = zero4; acc_w2q[j][i] = zero4; acc_w1k[i][j] = zero4; acc_w2k[j][i] = zero4; }
#pragma unroll
        for (int r = 0; r < 4; ++r) { agq[i][r] = abq[i][r] = agk[i][r] = abk[i][r] = agl[i][r] = abl[i][r] = 0.f; }
    }

    // this wave's 16 rows of every row buffer
    const int row0 = 16 * wave;
    float* my_q = sq + (size_t)(row0 + n) * LD + g4;   // D-layout row access (float4 per t)
    float* my_k = sk + (size_t)(row0 + n) * LD + g4;
    float* my_v = sv + (size_t)(row0 + n) * LD + g4;
    float* my_o = so + (size_t)(row0 + n) * LD + g4;
    float* my_g = sg + (size_t)(row0 + n) * LD + g4;
    const float* wg_q = sq + (size_t)(row0 + g4) * LD + n;   // token-contraction access: lane group g reads rows 4g + ks
    const float* wg_v = sv + (size_t)(row0 + g4) * LD + n;
    const float* wg_o = so + (size_t)(row0 + g4) * LD + n;
    const float* wg_g = sg + (size_t)(row0 + g4) * LD + n;

    // output: [G][CSZ] scenario-independent part, then [G + S][TSZ] generated-weight records (record index =
    // workgroup + scenario: strictly increasing along the global tile list, hence unique)
    constexpr int CSZ = 4 * D * D + 6 * D, TSZ = 4 * D * U;
    float* common = slabs + (size_t)blockIdx.x * CSZ;
    float* records = slabs + (size_t)gridDim.x * CSZ;
    float* stage = sq;   // the five row buffers are contiguous: 5 * 64 * LD floats of staging space
    using KTc = std::integral_constant<int, KT>;
    using UTc = std::integral_constant<int, UT>;
    // combine the four waves' accumulators of one matrix through LDS in a fixed order, write it out, clear it
    auto flush = [&](auto& acc, auto mtc, auto ntc, float* dst, bool live) {
        constexpr int MT_ = decltype(mtc)::value, NT_ = decltype(ntc)::value;
        constexpr int ncols = 16 * NT_, sz = MT_ * 16 * ncols;
        if (live) {
#pragma unroll
            for (int mt = 0; mt < MT_; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT_; ++nt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stage[wave * sz + (16 * mt + g4 + r) * ncols + 16 * nt + n] = acc[mt][nt][r];
                    acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < sz; e += kFusedBlock)
            dst[e] = live ? ((stage[e] + stage[sz + e]) + stage[2 * sz + e]) + stage[3 * sz + e] : 0.f;
        __syncthreads();
    };

    STAMP_DECL
    STAMP(8);
    int pre = 0;
    for (int scen = 0; scen < a.S && pre < wr.g1; ++scen) {
      const int nt_s = tiles_of(a.seg, scen, Tsamp);
      const int t0 = max(wr.g0, pre) - pre, t1 = min(wr.g1, pre + nt_s) - pre;
      pre += nt_s;
      if (t0 >= t1) continue;
      // ---- this scenario's generated MetaNet weights, both orientations (previous tile loop ended on a barrier) -----
      if (gate) {
          for (int i = threadIdx.x; i < D; i += blockDim.x) {
              if (meta_q) lnq_g[i] = 2.0f * a.tab_q[(size_t)scen * a.tab_stride + i];
              if (meta_k) lnk_g[i] = 2.0f * a.tab_k[(size_t)scen * a.tab_stride + i];
          }
      }
      if (bilin) {                                   // block-diagonal image: wb[h d + i][h d + j] = M[scenario, h][i][j]
          const float* row = a.tab_q + (size_t)scen * a.tab_stride;
          for (int e = threadIdx.x; e < D * D; e += blockDim.x) {
              const int r = e / D, c = e - r * D;
              w1q[r * LD + (TR ? c : c ^ img_flip(r))] = (r / d == c / d) ? row[(r / d) * d * d + (r % d) * d + (c % d)] : 0.f;
          }
      }
      if (mlp_q) {
          const float* row = a.tab_q + (size_t)scen * a.tab_stride;
          {
              const ImageJob jobs[2] = {{row, w1q, D, U, LU, false}, {row + D * U, w2q, U, D, LD, false}};
              stage_image_batch<2, (D * U / 4 + kFusedBlock - 1) / kFusedBlock, !TR>(jobs);
          }
          if constexpr (TR) {
              stage_image(row, w1qT, D, U, LD, true);
              stage_image(row + D * U, w2qT, U, D, LU, true);
          }
      }
      if (mlp_k && (!same_tab || !mlp_q)) {
          const float* row = a.tab_k + (size_t)scen * a.tab_stride;
          {
              const ImageJob jobs[2] = {{row, w1k, D, U, LU, false}, {row + D * U, w2k, U, D, LD, false}};
              stage_image_batch<2, (D * U / 4 + kFusedBlock - 1) / kFusedBlock, !TR>(jobs);
          }
          if constexpr (TR) {
              stage_image(row, w1kT, D, U, LD, true);
              stage_image(row + D * U, w2kT, U, D, LU, true);
          }
      }
      __syncthreads();
      const int lo = a.seg[scen], hi = a.seg[scen + 1];
      // With one wave per SIMD nothing hides a global load that is waited for where it is issued, and the counter the waits use
      // (vmcnt) retires loads IN ORDER - a wait for a young load also waits for every older one.  The loads of a tile are therefore
      // spread over the PREVIOUS tile, each issued where the ones before it have long landed, all of them unconditionally (a
      // load inside a divergent branch makes the compiler flush the counter where the branch rejoins):
      //   top of tile t      sample indices of tile t + 1 (token lane: b_next; attention task lane: tb_next)
      //   after phase A      the upstream gradient rows dy of tile t (consumed in phase C)
      //   after phase C      row id of the input row of tile t + 1 (first layer with the gather fused in: x_rows[b f])
      //   start of phase F   the input row of tile t + 1 and what the forward saved of its MetaNet (a whole phase of products ahead
      //                      of their use at the top of that tile; at the END of phase F - round 3 - the top of the tile waited)
      // Tile t1 - 1 "prefetches" itself again (clamped index): a few wasted loads instead of a branch.
      const int tok = row0 + n;
      const int ls_tok = tok / F, f_tok = tok - ls_tok * F;      // this lane's (sample, field) inside any tile that holds it
      float x_next[KT][4];
      auto sample_of = [&](int tile_) {
          const int first_ = lo + tile_ * Tsamp;
          return a.order[first_ + (tok < min(Tsamp, hi - first_) * F ? ls_tok : 0)];
      };
      auto task_sample_of = [&](int tile_) {      // (the sample of this lane's attention row in phase B: its dropout key)
          const int first_ = lo + tile_ * Tsamp;
          return a.order[first_ + min(MFA ? m_ls_raw : t0_ls, min(Tsamp, hi - first_) - 1)];
      };
      // row of a.x that holds this lane's input row: the position b F + f itself, or - gather fused in - the id stored there
      // (the raw loaded id travels to the point of use: any arithmetic on it right here - a sign extension - would be a wait
      // right here.  B F < 2^31: checked by the launcher)
      auto row_of = [&](int tile_, int b_) -> int {
          const int first_ = lo + tile_ * Tsamp;
          const int at = b_ * F + (tok < min(Tsamp, hi - first_) * F ? f_tok : 0);
          return a.x_rows ? a.x_rows[at] : at;
      };
      // (saved attention: the tile's dy rows are copied straight into LDS, 16 bytes per lane and KT rounds; lane c of round k
      // copies from sample (c + 256 k) / (F D / 4) of the tile - its index, too, is fetched a tile ahead)
      const int per4 = F * D / 4;
      int cb_next[KT];
      auto copy_samples_of = [&](int tile_) {
          const int first_ = lo + tile_ * Tsamp;
          const int last_ = min(Tsamp, hi - first_) - 1;
#pragma unroll
          for (int k_ = 0; k_ < KT; ++k_) cb_next[k_] = a.order[first_ + min(((int)threadIdx.x + kFusedBlock * k_) / per4, last_)];
      };
      // The hand-over of a tile (saved attention: softmax numerators, 1 / sum, keep words into their caches; the tile's dy rows
      // [sample][F][D] and saved attention outputs, unpadded, into the dS cache, which phase C reads once) is copied straight
      // into LDS, 16 bytes per lane, at the TOP of the tile (the caches are dead since the barrier behind the previous phase E)
      // and waited for at the end of phase A; the first tile of a scenario range: in front of the loop.
      // (Round 6 tried the copy for tile t + 1 at the start of tile t's phase F, a whole phase ahead of its use: +7.5 us per
      //  launch, 186 against 178.5 - an LDS-DMA piece costs 100-185 cycles to issue inside a phase that is busy with matrix
      //  products and LDS reads, eleven pieces per wave and tile; at the top of the tile nothing else is in flight.)
      auto handover_copy = [&](int tile_) {
          const int first_ = lo + tile_ * Tsamp;
          const int nS_ = min(Tsamp, hi - first_);
#pragma unroll
          for (int k_ = 0; k_ < KT; ++k_) {
              const int c = (int)threadIdx.x + kFusedBlock * k_;
              const int ls_ = c / per4;
              if (c < nS_ * per4) {
                  __builtin_amdgcn_global_load_lds(dy + (size_t)cb_next[k_] * F * D + 4 * (c - ls_ * per4),
                                                   (lds_ptr)(sDS + 4 * (kFusedBlock * k_ + 64 * wave)), 16, 0, 0);
                  __builtin_amdgcn_global_load_lds(save_o + (size_t)first_ * F * D + 4 * c,
                                                   (lds_ptr)(sDS + Tsamp * F * D + 4 * (kFusedBlock * k_ + 64 * wave)), 16, 0, 0);
              }
          }
          const int n4 = nS_ * F * HF / 4;
          const float* gp = a.attn_save + (size_t)first_ * F * HF;
          for (int c0 = 0; c0 < n4; c0 += kFusedBlock) {
              const int c = c0 + (int)threadIdx.x;
              if (c < n4) __builtin_amdgcn_global_load_lds(gp + 4 * c, (lds_ptr)(sP + 4 * (c0 + 64 * wave)), 16, 0, 0);
          }
          if ((int)threadIdx.x < nS_ * HF) {
              __builtin_amdgcn_global_load_lds(save_inv + (size_t)first_ * HF + threadIdx.x, (lds_ptr)(st_inv + 64 * wave), 4, 0, 0);
              __builtin_amdgcn_global_load_lds(save_keep + (size_t)first_ * HF + threadIdx.x, (lds_ptr)(st_keep + 64 * wave), 4, 0, 0);
          }
      };
      int b_next = sample_of(t0), tb_next = task_sample_of(t0);
      if (has_save && !REGH) {
          copy_samples_of(t0);
          handover_copy(t0);      // (the caches are dead: the previous range ended on the barrier behind its last phase F)
      }
      int xrow_next = row_of(t0, b_next);
      load_frag<KT>(a.x + (size_t)xrow_next * D + g4, x_next);
      // saved MetaNet rows: zh of both roles and their 1 / std of this lane's token, straight by sorted position - fetched where
      // the input row is, a tile ahead (padding lanes read the tile's first token)
      float zq_next[KT][4], zk_next[KT][4], rq_next = 0.f, rk_next = 0.f;
      auto fetch_z = [&](int tile_) {
          const int first_ = lo + tile_ * Tsamp;
          const size_t pos = (size_t)first_ * F + (tok < min(Tsamp, hi - first_) * F ? tok : 0);
          load_frag<KT>(save_z + pos * 2 * D + g4, zq_next);
          load_frag<KT>(save_z + pos * 2 * D + D + g4, zk_next);
          rq_next = save_r[pos * 2];
          rk_next = save_r[pos * 2 + 1];
      };
      if (has_zsave) fetch_z(t0);
      for (int tile = t0; tile < t1; ++tile) {
        const int first = lo + tile * Tsamp;
        const int nS = min(Tsamp, hi - first), ntok = nS * F, ntt = (ntok + 15) >> 4;
        const bool has_tile = wave < ntt;
        const bool valid = has_tile && tok < ntok;
        const int f = valid ? f_tok : 0;
        const int b = b_next, tb0 = tb_next;
        const int tile_n = min(tile + 1, t1 - 1);
        b_next = sample_of(tile_n);
        tb_next = task_sample_of(tile_n);
        // (saved attention: this tile's hand-over, see handover_copy; the sample indices of the NEXT tile's dy rows are fetched
        //  here, a tile ahead of the copy that needs them)
        if (has_save && !REGH && tile > t0) handover_copy(tile);
        if (has_save && !REGH) copy_samples_of(tile_n);
        // REGH: this token's upstream gradient row and saved attention output, consumed in phase C (padding lanes read the tile's
        // first token and are masked where the rows are consumed)
        float gy_reg[KT][4], o_reg[KT][4];
        if constexpr (REGH) {
            load_frag<KT>(dy + ((size_t)b * F + f) * D + g4, gy_reg);
            load_frag<KT>(save_o + ((size_t)first * F + (valid ? tok : 0)) * D + g4, o_reg);
        }
        // HEADF: the sample's label and dense columns, consumed in phase C - issued here, raw (see the load pipeline above)
        float label_pre = 0.f, dense_pre[kHeadDenseMax] = {0.f, 0.f};
        if constexpr (HEADF) {
            label_pre = hd.labels[b];
#pragma unroll
            for (int j = 0; j < kHeadDenseMax; ++j) dense_pre[j] = hd.dense[(size_t)b * hd.dense_stride + hd.dense_col[j]];
        }
        const uint32_t key_q = drop_sample_key(dc.key[kSiteMetaQ], (uint32_t)b);
        const uint32_t key_k = drop_sample_key(dc.key[kSiteMetaK], (uint32_t)b);
        const uint32_t key_o = drop_sample_key(dc.key[kSiteOut], (uint32_t)b);

        // token-wise state that lives from phase A to phase F
        float x[KT][4], q0[KT][4], k0[KT][4], hq[UT][4], hk[UT][4], zhq[KT][4], zhk[KT][4], dr[KT][4];
        float rstd_q = 0.f, rstd_k = 0.f;
        // keep flags of this token lane at the MetaNet-Q / MetaNet-K / output sites (bits 0-7 / 8-15 / 16-23), generated once
        uint32_t keepbits = 0xFFFFFFFFu;
        if (dc.on && has_tile)
            keepbits = token_keep_bits<KT>(key_q, f, D, g4, dc.thresh) | (token_keep_bits<KT>(key_k, f, D, g4, dc.thresh) << 8) |
                       (token_keep_bits<KT>(key_o, f, D, g4, dc.thresh) << 16);

        STAMP(0);
        // out = in x weight along the forward direction of an image; the MetaNet's hidden rows relu(in W1)
        auto fwd_w2 = [&](float* img, const float (&in_)[UT][4], float (&out_)[KT][4]) {
            chain<UT, KT, LD>(img + lo_d, in_, out_);
        };
        auto hidden = [&](float* img, const float (&in_)[KT][4], float (&out_)[UT][4]) {
            chain<KT, UT, LU>(img + lo_u, in_, out_);
#pragma unroll
            for (int t = 0; t < UT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) out_[t][r] = fmaxf(out_[t][r], 0.f);
        };
        // ================= phase A: forward chain ====================================================================
        if (has_tile) {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[t][r] = x_next[t][r];
            float v[KT][4], q[KT][4], k[KT][4];
            chain<KT, KT, LD>(wq + lo_d, x, q0);
            chain<KT, KT, LD>(wk + lo_d, x, k0);
            chain<KT, KT, LD>(wv + lo_d, x, v);
            if (mlp_q) {
                // (saved rows: the hidden rows are only needed by phase F, which computes them itself - 2 x 16 registers less to
                //  carry through the attention phases)
                if (!has_zsave) hidden(w1q, q0, hq);
                if (has_zsave) {      // the forward of this step left the normalised rows: no W2 product, no statistics
#pragma unroll
                    for (int t = 0; t < KT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) zhq[t][r] = zq_next[t][r];
                    rstd_q = rq_next;
                } else {
                float m[KT][4];
                fwd_w2(w2q, hq, m);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float mm = m[t][r];
                        if (dc.on) mm = (keepbits >> (4 * t + r)) & 1u ? mm * dc.scale : 0.f;
                        m[t][r] = mm + q0[t][r];
                    }
                layer_norm_keep<KT>(m, zhq, rstd_q);
                }
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const float4 gg = *reinterpret_cast<const float4*>(lnq_g + 16 * t + g4);
                    const float4 bb = *reinterpret_cast<const float4*>(lnq_b + 16 * t + g4);
                    q[t][0] = zhq[t][0] * gg.x + bb.x; q[t][1] = zhq[t][1] * gg.y + bb.y;
                    q[t][2] = zhq[t][2] * gg.z + bb.z; q[t][3] = zhq[t][3] * gg.w + bb.w;
                }
            } else if (bilin) {
                chain<KT, KT, LD>(w1q + lo_d, q0, q);     // q_h = q0_h M[s, h]
            } else {
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const float4 vq = *reinterpret_cast<const float4*>(lnq_g + 16 * t + g4);
                    const float gv_[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) q[t][r] = (gate && meta_q) ? q0[t][r] * gv_[r] : q0[t][r];
                }
            }
            if (mlp_k) {
                if (!has_zsave) hidden(w1k, k0, hk);
                if (has_zsave) {
#pragma unroll
                    for (int t = 0; t < KT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) zhk[t][r] = zk_next[t][r];
                    rstd_k = rk_next;
                } else {
                float m[KT][4];
                fwd_w2(w2k, hk, m);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float mm = m[t][r];
                        if (dc.on) mm = (keepbits >> (8 + 4 * t + r)) & 1u ? mm * dc.scale : 0.f;
                        m[t][r] = mm + k0[t][r];
                    }
                layer_norm_keep<KT>(m, zhk, rstd_k);
                }
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const float4 gg = *reinterpret_cast<const float4*>(lnk_g + 16 * t + g4);
                    const float4 bb = *reinterpret_cast<const float4*>(lnk_b + 16 * t + g4);
                    k[t][0] = zhk[t][0] * gg.x + bb.x; k[t][1] = zhk[t][1] * gg.y + bb.y;
                    k[t][2] = zhk[t][2] * gg.z + bb.z; k[t][3] = zhk[t][3] * gg.w + bb.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const float4 vk = *reinterpret_cast<const float4*>(lnk_g + 16 * t + g4);
                    const float gv_[4] = {vk.x, vk.y, vk.z, vk.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) k[t][r] = (gate && meta_k) ? k0[t][r] * gv_[r] : k0[t][r];
                }
            }
            store_frag<KT>(my_q, q);
            store_frag<KT>(my_k, k);
            store_frag<KT>(my_v, v);
        }
        if (has_save && !REGH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the caches have landed in LDS
        lds_barrier();

        STAMP(1);
        // REGH: this attention lane's softmax numerators (own query row, all keys), 1 / sum and keep word: issued here, a phase C
        // ahead of phase D (a lane of a sample the tile does not have reads the tile's last sample: its results are not stored)
        float p_reg[FT ? FP : 1], inv_reg = 0.f;
        uint32_t keep_reg = 0;
        if constexpr (REGH) {
            const int tls_g = min(m_ls_raw, nS - 1), iq_g = min(m_i, FT - 1);
            const float* pb = a.attn_save + (((size_t)(first + tls_g) * H + m_h) * FT + iq_g) * FP;      // [p][H][F][FP]: this lane's row
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) {
                const float4 t4 = *reinterpret_cast<const float4*>(pb + 4 * jb);
                p_reg[4 * jb] = t4.x; p_reg[4 * jb + 1] = t4.y; p_reg[4 * jb + 2] = t4.z; p_reg[4 * jb + 3] = t4.w;
            }
            inv_reg = save_inv[(size_t)(first + tls_g) * HF + m_h * FT + iq_g];
            keep_reg = __float_as_uint(save_keep[(size_t)(first + tls_g) * HF + m_h * FT + iq_g]);
        }
        // the upstream gradient rows of phase C: issued here, a phase ahead (HBM latency under the attention forward); padding
        // lanes read a real row (sample 0 of the tile, field 0) and are masked where the rows are consumed
        float gy_pre[KT][4];
        if constexpr (!HEADF)
            if (!has_save) load_frag<KT>(dy + ((size_t)b * F + f) * D + g4, gy_pre);
        // ================= phase B: attention forward; cache numerators, 1/sum and dropout keep bits ===================
        // Scores are staged in the task's row of the numerator cache (pre-scaled by log2(e)/sqrt(d)), keys in chunks of
        // four with all loads of a chunk issued before its results are stored; padding keys of the last chunk read the
        // last real row and are masked arithmetically.
        if constexpr ((kDiagSkip & 4) != 0) {
        } else if constexpr (MFA && !has_save) {
          // ---- matrix-pipe arm (see phase D): S^T blocks from the key rows picked by lane & 3 and the lane's own query row, the
          //      softmax lane-local on the 4 NJB accumulator registers, o = P V from the value feature pairs picked by lane & 3 ----
          if ((wave * SHW) / H < nS) {
            const int tls = min(m_ls_raw, Tsamp - 1), h = m_h;
            const int iq = min(m_i, FT - 1);
            const bool own = m_ls_raw < nS && m_i < FT && m_grp < SHW;
            const int told = tls * HF + h * FT + iq;
            float qe[d];
            {
                const float* qr = sq + (size_t)(tls * FT + iq) * LD + h * d;
                const float4 q0_ = *reinterpret_cast<const float4*>(qr), q1_ = *reinterpret_cast<const float4*>(qr + 4);
                qe[0] = q0_.x; qe[1] = q0_.y; qe[2] = q0_.z; qe[3] = q0_.w; qe[4] = q1_.x; qe[5] = q1_.y; qe[6] = q1_.z; qe[7] = q1_.w;
            }
            float ka[NJB][d];
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) {
                const float* kr = sk + (size_t)min(tls * FT + 4 * jb + m_sub, ROWS - 1) * LD + h * d;
                const float4 k0_ = *reinterpret_cast<const float4*>(kr), k1_ = *reinterpret_cast<const float4*>(kr + 4);
                ka[jb][0] = k0_.x; ka[jb][1] = k0_.y; ka[jb][2] = k0_.z; ka[jb][3] = k0_.w;
                ka[jb][4] = k1_.x; ka[jb][5] = k1_.y; ka[jb][6] = k1_.z; ka[jb][7] = k1_.w;
            }
            f32x2 vp[FT];
            {
                const float* vb = sv + (size_t)(tls * FT) * LD + h * d + 2 * m_sub;
#pragma unroll
                for (int j = 0; j < FT; ++j) vp[j] = ld_pair(vb + (size_t)j * LD);
            }
            f32x4 sc4[NJB];
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) sc4[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < d; ++e)
#pragma unroll
                for (int jb = 0; jb < NJB; ++jb) sc4[jb] = mfma_b16(ka[jb][e], qe[e], sc4[jb]);
            const float sc_scale = inv_sqrt_d * kLog2e;
            float ex[FT];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < FT; ++j) {
                ex[j] = sc4[j >> 2][j & 3] * sc_scale;
                mx = fmaxf(mx, ex[j]);
            }
            const uint32_t skey = drop_sample_key(dc.key[kSiteAttn], (uint32_t)tb0);
            const uint32_t block0 = drop_attn_elem(h, FT, iq, 0) >> 2;
            // (every lane writes: a padding / spare lane holds a copy of a real row - same values to the same addresses -, and
            //  the slots of a sample the tile does not have belong to nobody.  A branch around the stores would also make the
            //  compiler drain the LDS queue in front of every product that follows.)
            float* prow = sP + (size_t)tls * FT * HF + (h * FT + iq);
            float sum = 0.f;
            uint32_t keep = 0xFFFFFFFFu;
            f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) {
                const uint32_t kb = dc.on ? drop_keep4(skey, block0 + (uint32_t)jb, dc.thresh) : 0xFu;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 4 * jb + r;
                    if (j < FT) {
                        ex[j] = __builtin_amdgcn_exp2f(ex[j] - mx);
                        sum += ex[j];
                        float pe = ex[j];
                        if (dc.on) {
                            const bool kp = (kb >> r) & 1u;
                            pe = kp ? ex[j] * dc.scale : 0.f;
                            keep = kp ? keep : keep & ~(1u << j);
                        }
                        oa = mfma_b16(vp[j].x, pe, oa);
                        ob = mfma_b16(vp[j].y, pe, ob);
                        prow[j * HF] = ex[j];
                    }
                }
            }
            const float inv = 1.0f / sum;
            st_inv[told] = inv;
            st_keep[told] = keep;
            if (own) {
                float* orow = so + (size_t)(tls * FT + m_i) * LD + h * d;
                *reinterpret_cast<float4*>(orow) = make_float4(oa[0] * inv, ob[0] * inv, oa[1] * inv, ob[1] * inv);
                *reinterpret_cast<float4*>(orow + 4) = make_float4(oa[2] * inv, ob[2] * inv, oa[3] * inv, ob[3] * inv);
            }
          }
        } else
        if (const int task = threadIdx.x; !has_save && task < nS * H * F) {      // (at most 64 H <= 256 tasks per tile: one per thread)
            const int tls = t0_ls, h = t0_h, i = t0_i;
            const int tb = tb0;
            f32x2 qi[d / 2];
            load_row<d>(sq + (size_t)(tls * F + i) * LD + h * d, qi);
            const float* kbase = sk + (size_t)(tls * F) * LD + h * d;
            const float* vbase = sv + (size_t)(tls * F) * LD + h * d;
            float* prow = sP + (size_t)tls * F * HF + (h * F + i);       // numerator of key j at prow[j HF]: [sample][key][task]
            const float sc_scale = inv_sqrt_d * kLog2e;
            float mx = -INFINITY;
            auto chunk1 = [&](const int j0) {
                f32x2 kr[4][d / 2];
                float sc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) load_row<d>(kbase + (size_t)min(j0 + u, F - 1) * LD, kr[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    sc[u] = dot_row<d>(qi, kr[u]) * sc_scale;
                    mx = fmaxf(mx, sc[u]);          // a padding key repeats the last real score: the maximum is unchanged
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j0 + u < F) prow[(j0 + u) * HF] = sc[u];
            };
            ATTN_CHUNKS(j0, chunk1, true);
            f32x2 oacc[d / 2];
#pragma unroll
            for (int e = 0; e < d / 2; ++e) oacc[e] = f32x2{0.f, 0.f};
            float sum = 0.f;
            uint32_t keep = 0xFFFFFFFFu;
            const uint32_t skey = drop_sample_key(dc.key[kSiteAttn], (uint32_t)tb);
            const uint32_t block0 = drop_attn_elem(h, F, i, 0) >> 2;
            // `guarded`: the chunk may hold keys beyond F (the last, partial chunk).  With the field count a constant the whole
            // chunks run as a rolled loop WITHOUT guards - unrolled, this loop alone spills 56 VGPRs; rolled with guards, hipcc
            // turns every `j < F` into a scalar branch around its v_exp (eight taken-or-not branches per iteration at one wave
            // per SIMD) - and the partial chunk follows as straight-line code with constant guards.
            auto chunk2 = [&](const int j0, auto guarded_c) {
                constexpr bool guarded = decltype(guarded_c)::value;
                f32x2 vr[4][d / 2];
                float ex[4];
                const uint32_t kb = dc.on ? drop_keep4(skey, block0 + (uint32_t)(j0 >> 2), dc.thresh) : 0xFu;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = guarded ? min(j0 + u, F - 1) : j0 + u;
                    ex[u] = prow[j * HF];
                    load_row<d>(vbase + (size_t)j * LD, vr[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u;
                    ex[u] = __builtin_amdgcn_exp2f(ex[u] - mx);
                    if (guarded && j >= F) ex[u] = 0.f;
                    sum += ex[u];
                    float pe = ex[u];
                    if (dc.on) {
                        const bool kp = (kb >> u) & 1u;
                        pe = kp ? ex[u] * dc.scale : 0.f;
                        keep = kp ? keep : keep & ~(1u << (j & 31));
                    }
                    axpy_row<d>(pe, vr[u], oacc);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (!guarded || j0 + u < F) prow[(j0 + u) * HF] = ex[u];
            };
            if constexpr (FT != 0) {
                constexpr int kWhole = FT & ~3;
#pragma unroll 1
                for (int j0 = 0; j0 < kWhole; j0 += 4) chunk2(j0, std::false_type{});
                if constexpr (kWhole < FT) chunk2(kWhole, std::true_type{});
            } else {
#pragma unroll 1
                for (int j0 = 0; j0 < F; j0 += 4) chunk2(j0, std::true_type{});
            }
            const float inv = 1.0f / sum;
            st_inv[task] = inv;
            st_keep[task] = keep;
            store_row<d>(so + (size_t)(tls * F + i) * LD + h * d, oacc, inv);
        }
        lds_barrier();

        STAMP(2);
        // ================= phase C: output block forward + backward ======================================================
        // out = in x (D x D weight), along the forward direction of the image `f_` or the backward direction (image `b_`)
        auto prod_dd = [&](float* img, bool back_, const float (&in_)[KT][4], float (&out_)[KT][4]) {
            if (!back_) {
                chain<KT, KT, LD>(img + lo_d, in_, out_);
            } else {
                if constexpr (TR) chain<KT, KT, LD>(img + lo_d, in_, out_);
                else chain_t<KT, KT, LD>(img + lt_d, in_, out_);
            }
        };
        float zh[KT][4], gy[KT][4], keep[KT][4];      // normalised output rows, upstream gradient, dropout x ReLU factor of du
        float rstd_o = 0.f;
        const float* staged = sDS + ((size_t)(valid ? ls_tok : 0) * F + f) * D + g4;      // (saved attention: this token's dy row)
        if (has_tile) {
            float o[KT][4], u[KT][4];
            if constexpr (REGH) {
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][r] = valid ? o_reg[t][r] : 0.f;
                store_frag<KT>(my_o, o);          // (dWo below reads the wave's o rows from LDS)
            } else if (has_save) {
                load_frag<KT>(staged + Tsamp * F * D, o, valid);
                store_frag<KT>(my_o, o);          // (dWo below reads the wave's o rows from LDS)
            } else {
                load_frag<KT>(my_o, o);
            }
            prod_dd(woT, false, o, u);
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float val = u[t][r], kf = 1.0f;
                    if (relu_out) { kf = val > 0.f ? 1.0f : 0.f; val = fmaxf(val, 0.f); }
                    if (dc.on) {
                        const float mk = (keepbits >> (16 + 4 * t + r)) & 1u ? dc.scale : 0.f;
                        val *= mk; kf *= mk;
                    }
                    keep[t][r] = kf;
                    u[t][r] = use_res ? val + x[t][r] : val;
                }
            layer_norm_keep<KT>(u, zh, rstd_o);
        }
        if constexpr (HEADF) {
            // ---- the head on the rows this tile just recomputed: y = zh gamma + beta is the layer's OUTPUT (satrans.py:98);
            //      logit_b = sum over the sample's F tokens of y . w_head[f] (+ dense columns + bias), satrans.py:244-255 ----
            const int ls_c = valid ? ls_tok : 0;
            if (has_tile) {
                float pd = 0.f;
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const float4 gg = *reinterpret_cast<const float4*>(ln_g + 16 * t + g4);
                    const float4 bb = *reinterpret_cast<const float4*>(ln_b + 16 * t + g4);
                    const float4 ww = *reinterpret_cast<const float4*>(s_wh + f * D + 16 * t + g4);
                    pd = fmaf(zh[t][0] * gg.x + bb.x, ww.x, pd); pd = fmaf(zh[t][1] * gg.y + bb.y, ww.y, pd);
                    pd = fmaf(zh[t][2] * gg.z + bb.z, ww.z, pd); pd = fmaf(zh[t][3] * gg.w + bb.w, ww.w, pd);
                }
                pd = token_sum(pd);
                if (g == 0) s_dot[row0 + n] = valid ? pd : 0.f;
                if (hd.n_dense > 0 && valid && f == 0 && g == 0) {
                    float dn = 0.f;
#pragma unroll
                    for (int j = 0; j < kHeadDenseMax; ++j)
                        if (j < hd.n_dense) dn = fmaf(dense_pre[j], s_wh[F * D + j], dn);
                    s_dn[ls_c] = dn;
                }
            }
            lds_barrier();
            if (has_tile) {
                float z = 0.f;
                if constexpr (FT != 0) {
#pragma unroll
                    for (int f2 = 0; f2 < FT; ++f2) z += s_dot[ls_c * FT + f2];
                } else {
                    for (int f2 = 0; f2 < F; ++f2) z += s_dot[ls_c * F + f2];
                }
                z = (z + s_dn[ls_c]) + head_bias;
                const float pr = 1.0f / (1.0f + expf(-z));
                const float tl = label_pre;
                const float pq = (1.0f - pr) * pr;
                float dl, lossv;
                if (hd.loss_kind == SATRANS_LOSS_MSE) {            // F.mse_loss(reduction='sum')
                    lossv = (pr - tl) * (pr - tl);
                    dl = 2.0f * (pr - tl) * pq;
                } else if (hd.loss_kind == SATRANS_LOSS_MAE) {     // F.l1_loss(reduction='sum')
                    lossv = fabsf(pr - tl);
                    dl = (pr > tl ? 1.0f : (pr < tl ? -1.0f : 0.0f)) * pq;
                } else {                                           // binary_cross_entropy: both logs clamped at -100 (as head_kernel)
                    const float lp = fmaxf(logf(pr), -100.f), lq = fmaxf(logf(1.0f - pr), -100.f);
                    lossv = -(tl * lp + (1.0f - tl) * lq);
                    dl = (pr - tl) / fmaxf(pq, 1e-12f) * pq;
                }
                if (!valid) dl = 0.f;
                if (valid && f == 0 && g == 0) {                   // one lane per sample: outputs and the per-workgroup sums
                    hd.prob[b] = pr;
                    if (hd.logit) hd.logit[b] = z;
                    head_loss += lossv;
                    head_gb += dl;
#pragma unroll
                    for (int j = 0; j < kHeadDenseMax; ++j)
                        if (j < hd.n_dense) s_hwd[ls_c * kHeadDenseMax + j] = fmaf(dl, dense_pre[j], s_hwd[ls_c * kHeadDenseMax + j]);
                }
                // dy = dlogit w_head[f] (never leaves the registers); dW_head[f] += dlogit y in this lane's own LDS slots
                float* hw = s_hw + (size_t)threadIdx.x * 4 * KT;
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const float4 gg = *reinterpret_cast<const float4*>(ln_g + 16 * t + g4);
                    const float4 bb = *reinterpret_cast<const float4*>(ln_b + 16 * t + g4);
                    const float4 ww = *reinterpret_cast<const float4*>(s_wh + f * D + 16 * t + g4);
                    float4 ac = *reinterpret_cast<float4*>(hw + 4 * t);
                    ac.x = fmaf(dl, zh[t][0] * gg.x + bb.x, ac.x); ac.y = fmaf(dl, zh[t][1] * gg.y + bb.y, ac.y);
                    ac.z = fmaf(dl, zh[t][2] * gg.z + bb.z, ac.z); ac.w = fmaf(dl, zh[t][3] * gg.w + bb.w, ac.w);
                    *reinterpret_cast<float4*>(hw + 4 * t) = ac;
                    gy[t][0] = dl * ww.x; gy[t][1] = dl * ww.y; gy[t][2] = dl * ww.z; gy[t][3] = dl * ww.w;
                }
            }
        } else if (has_tile) {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) gy[t][r] = valid ? (REGH ? gy_reg[t][r] : gy_pre[t][r]) : 0.f;
            if (has_save && !REGH) load_frag<KT>(staged, gy, valid);
        }
        if (has_tile) {
            layer_norm_bwd<KT>(gy, zh, rstd_o, ln_g, g4, agl, abl);          // gy is now dr
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dr[t][r] = use_res ? gy[t][r] : 0.f;
                    gy[t][r] *= keep[t][r];                                     // du
                }
            store_frag<KT>(my_g, gy);                                           // du rows (zero for padding tokens)
            WGRAD<KT, KT, 0, 0, LD, LD>(wg_g, wg_o, acc_wo);                    // dWo[o][i] += du^T o
            float go[KT][4];
            prod_dd(wo, true, gy, go);                                          // go = du Wo
            store_frag<KT>(my_o, go);
        }
        lds_barrier();

        STAMP(3);
        xrow_next = row_of(tile_n, b_next);
        // ================= phase D: softmax backward by rows: dS_ij (cached for phase E) and dq_i =========================
        // pass 1: dP_ij = (go_i . v_j) * mask_ij staged in the task's row of the dS cache, dot_i = sum_j P_ij dP_ij;
        // pass 2: dS_ij = P_ij (dP_ij - dot_i) / sqrt(d) replaces it, the numerator cache row becomes P_ij * mask_ij
        //         (the coefficient of dv_j), dq_i = sum_j dS_ij k_j.
        if constexpr ((kDiagSkip & 1) != 0) {
        } else if constexpr (MFA) {
          // ---- matrix-pipe arm.  Every lane of a wave that holds a sample of the tile runs the products (the A operands are
          //      shared rows addressed by lane & 3, whatever the lane's own row). ---------------------------------------------
          if ((wave * SHW) / H < nS) {
            const int tls = min(m_ls_raw, Tsamp - 1), h = m_h;
            const int iq = min(m_i, FT - 1);                       // a padding row computes a copy of the last real row
            const bool own = m_ls_raw < nS && m_i < FT && m_grp < SHW;
            const int told = tls * HF + h * FT + iq;                // this row in the forward's (unpadded) task numbering
            // dP^T block jb: register r = dP[i][4 jb + r] = sum_e v[4 jb + r][e] go[i][e]
            float ge[d];
            {
                const float* gr = so + (size_t)(tls * FT + iq) * LD + h * d;
                const float4 g0 = *reinterpret_cast<const float4*>(gr), g1 = *reinterpret_cast<const float4*>(gr + 4);
                ge[0] = g0.x; ge[1] = g0.y; ge[2] = g0.z; ge[3] = g0.w; ge[4] = g1.x; ge[5] = g1.y; ge[6] = g1.z; ge[7] = g1.w;
            }
            float va[NJB][d];
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) {
                const float* vr = sv + (size_t)min(tls * FT + 4 * jb + m_sub, ROWS - 1) * LD + h * d;
                const float4 v0 = *reinterpret_cast<const float4*>(vr), v1 = *reinterpret_cast<const float4*>(vr + 4);
                va[jb][0] = v0.x; va[jb][1] = v0.y; va[jb][2] = v0.z; va[jb][3] = v0.w;
                va[jb][4] = v1.x; va[jb][5] = v1.y; va[jb][6] = v1.z; va[jb][7] = v1.w;
            }
            float* prow = sP + (size_t)tls * FT * HF + (h * FT + iq);
            float pn[FT];
#pragma unroll
            for (int j = 0; j < FT; ++j) pn[j] = REGH ? p_reg[j] : prow[j * HF];
            const float inv = REGH ? inv_reg : st_inv[told];
            const uint32_t keep = REGH ? keep_reg : st_keep[told];
            const float scale = dc.scale;
            f32x4 dp[NJB];
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) dp[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < d; ++e)
#pragma unroll
                for (int jb = 0; jb < NJB; ++jb) dp[jb] = mfma_b16(va[jb][e], ge[e], dp[jb]);
            // the K feature pairs of the second product (lane & 3 picks the pair): in flight under the softmax arithmetic
            f32x2 kp[FT];
            {
                const float* kb = sk + (size_t)(tls * FT) * LD + h * d + 2 * m_sub;
#pragma unroll
                for (int j = 0; j < FT; ++j) kp[j] = ld_pair(kb + (size_t)j * LD);
            }
            float dsv[FT];
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < FT; ++j) {
                const float dpj = ((keep >> j) & 1u) ? dp[j >> 2][j & 3] * scale : 0.f;
                pn[j] *= inv;
                dot = fmaf(pn[j], dpj, dot);
                dsv[j] = dpj;
            }
            // dq[i][2 r], dq[i][2 r + 1] = sum_j k[j][h d + 2 r (+ 1)] dS[i][j]; dS and the masked probabilities go to the caches as
            // they are made (every lane stores, no branch: see phase B), between the products
            float* dsT = sDS + (size_t)((tls * H + h) * FP) * FP + m_i;        // dS[i][j] at dsT[j FP]
            f32x4 qa = {0.f, 0.f, 0.f, 0.f}, qb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < FT; ++j) {
                dsv[j] = pn[j] * (dsv[j] - dot) * inv_sqrt_d;
                const float pm = ((keep >> j) & 1u) ? pn[j] * scale : 0.f;
                qa = mfma_b16(kp[j].x, dsv[j], qa);
                qb = mfma_b16(kp[j].y, dsv[j], qb);
                dsT[j * FP] = dsv[j];
                prow[j * HF] = pm;
            }
            if (own) {
                float* qr = sg + (size_t)(tls * FT + m_i) * LD + h * d;
                *reinterpret_cast<float4*>(qr) = make_float4(qa[0], qb[0], qa[1], qb[1]);
                *reinterpret_cast<float4*>(qr + 4) = make_float4(qa[2], qb[2], qa[3], qb[3]);
            }
          }
        } else
        if (const int task = threadIdx.x; task < nS * H * F) {      // (at most 64 H <= 256 tasks per tile: one per thread)
            const int tls = t0_ls, h = t0_h, i = t0_i;
            f32x2 gi[d / 2];
            load_row<d>(so + (size_t)(tls * F + i) * LD + h * d, gi);
            const float* kbase = sk + (size_t)(tls * F) * LD + h * d;
            const float* vbase = sv + (size_t)(tls * F) * LD + h * d;
            float* prow = sP + (size_t)tls * F * HF + (h * F + i);
            float* drow = sDS + (size_t)task * F;
            const float inv = st_inv[task];
            const uint32_t keep = st_keep[task];
            const float scale = dc.scale;
            float dot = 0.f;
            struct Rows3 { f32x2 vr[4][d / 2]; float pj[4]; };
            auto load3 = [&](const int j0, Rows3& c) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = min(j0 + u, F - 1);
                    c.pj[u] = prow[j * HF];
                    load_row<d>(vbase + (size_t)j * LD, c.vr[u]);
                }
            };
            auto comp3 = [&](const int j0, Rows3& c) {
                float dp[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u;
                    dp[u] = dot_row<d>(gi, c.vr[u]);
                    dp[u] = ((keep >> (j & 31)) & 1u) ? dp[u] * scale : 0.f;
                    const float pj = j < F ? c.pj[u] * inv : 0.f;
                    dot = fmaf(pj, dp[u], dot);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j0 + u < F) drow[j0 + u] = dp[u];
            };
            auto chunk3 = [&](const int j0) {
                Rows3 c;
                load3(j0, c);
                comp3(j0, c);
            };
            if constexpr (kAttnPipe && FT != 0 && SAME) ATTN_PIPE(Rows3, load3, comp3)
            else ATTN_CHUNKS(j0, chunk3, true);
            f32x2 dq[d / 2];
#pragma unroll
            for (int e = 0; e < d / 2; ++e) dq[e] = f32x2{0.f, 0.f};
            struct Rows4 { f32x2 kr[4][d / 2]; float pj[4], ds[4]; };
            auto load4 = [&](const int j0, Rows4& c) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = min(j0 + u, F - 1);
                    c.pj[u] = prow[j * HF];
                    c.ds[u] = drow[j];
                    load_row<d>(kbase + (size_t)j * LD, c.kr[u]);
                }
            };
            auto comp4 = [&](const int j0, Rows4& c) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u;
                    c.pj[u] = j < F ? c.pj[u] * inv : 0.f;
                    c.ds[u] = c.pj[u] * (c.ds[u] - dot) * inv_sqrt_d;
                    c.pj[u] = ((keep >> (j & 31)) & 1u) ? c.pj[u] * scale : 0.f;
                    axpy_row<d>(c.ds[u], c.kr[u], dq);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j0 + u < F) { drow[j0 + u] = c.ds[u]; prow[(j0 + u) * HF] = c.pj[u]; }
            };
            auto chunk4 = [&](const int j0) {
                Rows4 c;
                load4(j0, c);
                comp4(j0, c);
            };
            // (pipelined: chunk c + 1 reads its own rows of the two caches, chunk c writes its own - disjoint)
            if constexpr (kAttnPipe && FT != 0 && SAME) ATTN_PIPE(Rows4, load4, comp4)
            else ATTN_CHUNKS(j0, chunk4, true);
            store_row<d>(sg + (size_t)(tls * F + i) * LD + h * d, dq, 1.0f);
        }
        // (matrix-pipe arm: phase E reads of the caches only what its own wave wrote in phase D, and rewrites of the k / v rows only
        //  the (sample, head) column slices its own wave has finished reading: a wave-local wait, no workgroup barrier)
        if constexpr (MFA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else lds_barrier();

        STAMP(4);
        // ================= phase E: by columns: dk_j = sum_i dS_ij q_i, dv_j = sum_i P_ij mask_ij go_i (in place of k_j, v_j)
        if constexpr ((kDiagSkip & 2) != 0) {
        } else if constexpr (MFA) {
          // ---- matrix-pipe arm: the lane's row is now KEY j; B = its own column of dS / of the masked probabilities (read as
          //      rows of the transposed caches), A = the q / go feature pairs picked by lane & 3 -------------------------------
          if ((wave * SHW) / H < nS) {
            const int tls = min(m_ls_raw, Tsamp - 1), h = m_h;
            const int jk = min(m_i, FT - 1);
            const bool own = m_ls_raw < nS && m_i < FT && m_grp < SHW;
            float dsc[FP], pmc[FT];
            f32x2 qp[FT], gp[FT];
            const float* qb_ = sq + (size_t)(tls * FT) * LD + h * d + 2 * m_sub;
            const float* gb_ = so + (size_t)(tls * FT) * LD + h * d + 2 * m_sub;
            // The LDS queue of a wave counts 15 reads at most and waits by count: of the 53 reads of this phase the first product
            // needs 24, and a wait for them is only expressible while fewer than 16 younger ones are in flight.  So: its own
            // operands + the 10 of the masked probabilities, half of the first product, the go pairs, its other half, the second
            // product - every wait a counted one, every read a product or more ahead of its use.
            {
                const float* dr_ = sDS + (size_t)((tls * H + h) * FP + jk) * FP;
#pragma unroll
                for (int ib = 0; ib < NJB; ++ib) {
                    const float4 t = *reinterpret_cast<const float4*>(dr_ + 4 * ib);
                    dsc[4 * ib] = t.x; dsc[4 * ib + 1] = t.y; dsc[4 * ib + 2] = t.z; dsc[4 * ib + 3] = t.w;
                }
#pragma unroll
                for (int i = 0; i < FT; ++i) qp[i] = ld_pair(qb_ + (size_t)i * LD);
            }
            {
                const float* pc = sP + (size_t)tls * FT * HF + (size_t)jk * HF + h * FT;
#pragma unroll
                for (int i = 0; i < FT; ++i) pmc[i] = pc[i];
            }
            f32x4 ka = {0.f, 0.f, 0.f, 0.f}, kb2 = {0.f, 0.f, 0.f, 0.f}, va2 = {0.f, 0.f, 0.f, 0.f}, vb2 = {0.f, 0.f, 0.f, 0.f};
            constexpr int IH = FT / 2;
#pragma unroll
            for (int i = 0; i < IH; ++i) {
                ka = mfma_b16(qp[i].x, dsc[i], ka);
                kb2 = mfma_b16(qp[i].y, dsc[i], kb2);
            }
#pragma unroll
            for (int i = 0; i < FT; ++i) gp[i] = ld_pair(gb_ + (size_t)i * LD);
#pragma unroll
            for (int i = IH; i < FT; ++i) {
                ka = mfma_b16(qp[i].x, dsc[i], ka);
                kb2 = mfma_b16(qp[i].y, dsc[i], kb2);
            }
#pragma unroll
            for (int i = 0; i < FT; ++i) {
                va2 = mfma_b16(gp[i].x, pmc[i], va2);
                vb2 = mfma_b16(gp[i].y, pmc[i], vb2);
            }
            if (own) {
                float* kr = sk + (size_t)(tls * FT + m_i) * LD + h * d;
                float* vr = sv + (size_t)(tls * FT + m_i) * LD + h * d;
                *reinterpret_cast<float4*>(kr) = make_float4(ka[0], kb2[0], ka[1], kb2[1]);
                *reinterpret_cast<float4*>(kr + 4) = make_float4(ka[2], kb2[2], ka[3], kb2[3]);
                *reinterpret_cast<float4*>(vr) = make_float4(va2[0], vb2[0], va2[1], vb2[1]);
                *reinterpret_cast<float4*>(vr + 4) = make_float4(va2[2], vb2[2], va2[3], vb2[3]);
            }
          }
        } else
        if (const int task = threadIdx.x; task < nS * H * F) {      // (at most 64 H <= 256 tasks per tile: one per thread)
            const int tls = t0_ls, h = t0_h, j = t0_i;
            f32x2 dk[d / 2], dv[d / 2];
#pragma unroll
            for (int e = 0; e < d / 2; ++e) { dk[e] = f32x2{0.f, 0.f}; dv[e] = f32x2{0.f, 0.f}; }
            const float* qbase = sq + (size_t)(tls * F) * LD + h * d;
            const float* gbase = so + (size_t)(tls * F) * LD + h * d;
            const float* dcol = sDS + (size_t)((tls * H + h) * F) * F + j;
            const float* pcol = sP + (size_t)tls * F * HF + (size_t)j * HF + h * F;      // the P of (query i, key j) at pcol[i]
            struct Rows5 { f32x2 qr[4][d / 2], gr[4][d / 2]; float ds[4], pm[4]; };
            auto load5 = [&](const int i0, Rows5& c) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = min(i0 + u, F - 1);
                    c.ds[u] = dcol[(size_t)i * F];
                    c.pm[u] = pcol[i];
                    load_row<d>(qbase + (size_t)i * LD, c.qr[u]);
                    load_row<d>(gbase + (size_t)i * LD, c.gr[u]);
                }
            };
            auto comp5 = [&](const int i0, Rows5& c) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool real = i0 + u < F;
                    axpy_row<d>(real ? c.ds[u] : 0.f, c.qr[u], dk);
                    axpy_row<d>(real ? c.pm[u] : 0.f, c.gr[u], dv);
                }
            };
            auto chunk5 = [&](const int i0) {
                Rows5 c;
                load5(i0, c);
                comp5(i0, c);
            };
            if constexpr (kAttnPipe && FT != 0 && SAME) ATTN_PIPE(Rows5, load5, comp5)
            else ATTN_CHUNKS(i0, chunk5, true);
            store_row<d>(sk + (size_t)(tls * F + j) * LD + h * d, dk, 1.0f);
            store_row<d>(sv + (size_t)(tls * F + j) * LD + h * d, dv, 1.0f);
        }
        lds_barrier();

        STAMP(5);
        // the next tile's input row (and what the forward saved of its MetaNet): issued HERE, a whole phase F ahead of their use at
        // the top of the next tile (at the end of phase F the dropout hashing of that top covered a few hundred cycles of an HBM
        // round trip, the rest was waited for)
        load_frag<KT>(a.x + (size_t)xrow_next * D + g4, x_next);
        if (has_zsave) fetch_z(tile_n);
        // ================= phase F: MetaNet and projection backward, weight gradients, dx ==================================
        if (has_tile) {
            float gq[KT][4], gk[KT][4];
            load_frag<KT>(my_g, gq, valid);      // gradient of the (post-MetaNet) queries
            load_frag<KT>(my_k, gk, valid);      // ... keys; the value gradients stay in sv for dWv and are read below
            // rows >= ntok of sv still hold forward values: they are neutralised by x = 0 in the dWv product and
            // masked when read as a fragment

            auto metanet_bwd = [&](float (&gout)[KT][4], const float (&zh)[KT][4], float rstd, const float* gam,
                                   float (&ag)[KT][4], float (&ab)[KT][4], int kshift, float (&h)[UT][4],
                                   const float (&in0)[KT][4], const float* w2T, const float* w1T, float* w1f,
                                   f32x4 (&acc_w1)[KT][UT], f32x4 (&acc_w2)[UT][KT]) {
                if (rehidden) hidden(w1f, in0, h);                              // (phase A skipped it, or its copy was let go)
                layer_norm_bwd<KT>(gout, zh, rstd, gam, g4, ag, ab);            // gout = dz
                float dm[KT][4];
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float mk = 1.0f;
                        if (dc.on) mk = (keepbits >> (kshift + 4 * t + r)) & 1u ? dc.scale : 0.f;
                        dm[t][r] = gout[t][r] * mk;
                    }
                // Order (round 6): a weight-gradient product reads back from LDS what this wave has just stored, and at one wave per
                // SIMD nothing hides that store -> load round trip - except the chain that does not depend on it.  So: store the
                // operands of dW2, run dh = dm W2^T (registers + weights only), THEN the dW2 products; store the operands of dW1,
                // run back = dh W1^T, THEN the dW1 products.  (-DSATRANS_EXP_WGRAD_INORDER: the products right behind their stores.)
#ifdef SATRANS_EXP_WGRAD_INORDER
                constexpr bool kLate = false;
#else
                constexpr bool kLate = true;
#endif
                // dW2[u][o] += h^T dm : h goes to the q (and o) rows of this tile, dm to the dq rows
                auto dw2_products = [&]() {
                    WGRAD<NB, KT, 0, 0, LD, LD>(wg_q, wg_g, acc_w2);
                    if constexpr (HB == 2) WGRAD<NB, KT, NB, 0, LD, LD>(wg_o, wg_g, acc_w2);
                };
                {
                    float part[KT][4];
#pragma unroll
                    for (int t = 0; t < KT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) part[t][r] = (t < NB) ? h[t % UT][r] : 0.f;
                    store_frag<KT>(my_q, part);
                    if constexpr (HB == 2) {
#pragma unroll
                        for (int t = 0; t < KT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) part[t][r] = (t < NB) ? h[(NB + t) % UT][r] : 0.f;
                        store_frag<KT>(my_o, part);
                    }
                    store_frag<KT>(my_g, dm);
                    if constexpr (!kLate) dw2_products();
                }
                // dh = (dm W2^T) * [h > 0]
                float dh[UT][4];
                if constexpr (TR) chain<KT, UT, LU>(w2T + lo_u, dm, dh);
                else chain_t<KT, UT, LD>(w2T + lt_d, dm, dh);              // w2T is then the forward image W2 [U][LD]
                if constexpr (kLate) dw2_products();
#pragma unroll
                for (int t = 0; t < UT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dh[t][r] = h[t][r] > 0.f ? dh[t][r] : 0.f;
                // dW1[i][u] += in0^T dh : in0 to the dq rows, dh to the q (and o) rows
                auto dw1_products = [&]() {
                    WGRAD<KT, NB, 0, 0, LD, LD>(wg_g, wg_q, acc_w1);
                    if constexpr (HB == 2) WGRAD<KT, NB, 0, NB, LD, LD>(wg_g, wg_o, acc_w1);
                };
                {
                    float part[KT][4];
#pragma unroll
                    for (int t = 0; t < KT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) part[t][r] = (t < NB) ? dh[t % UT][r] : 0.f;
                    store_frag<KT>(my_q, part);
                    if constexpr (HB == 2) {
#pragma unroll
                        for (int t = 0; t < KT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) part[t][r] = (t < NB) ? dh[(NB + t) % UT][r] : 0.f;
                        store_frag<KT>(my_o, part);
                    }
                    store_frag<KT>(my_g, in0, valid);
                    if constexpr (!kLate) dw1_products();
                }
                // gradient of the MetaNet input: dz + dh W1^T
                float back[KT][4];
                if constexpr (TR) chain<UT, KT, LD>(w1T + lo_d, dh, back);
                else chain_t<UT, KT, LU>(w1T + lt_u, dh, back);            // ... the forward image W1 [D][LU]
                if constexpr (kLate) dw1_products();
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gout[t][r] += back[t][r];
            };

            if (gate) {                         // q = q0 * 2 vec: d vec += 2 gq q0 (the 2 is applied at the flush), gq0 = gq * 2 vec
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const float4 vq = *reinterpret_cast<const float4*>(lnq_g + 16 * t + g4);
                    const float4 vk = *reinterpret_cast<const float4*>(lnk_g + 16 * t + g4);
                    const float gq_[4] = {vq.x, vq.y, vq.z, vq.w}, gk_[4] = {vk.x, vk.y, vk.z, vk.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (meta_q) { agq[t][r] = fmaf(gq[t][r], q0[t][r], agq[t][r]); gq[t][r] *= gq_[r]; }
                        if (meta_k) { agk[t][r] = fmaf(gk[t][r], k0[t][r], agk[t][r]); gk[t][r] *= gk_[r]; }
                    }
                }
            }
            if (bilin) {                        // q_h = q0_h M: dM += q0^T gq (block-diagonal part kept by the reducer), gq0 = gq M^T
                store_frag<KT>(my_q, q0, valid);
                store_frag<KT>(my_o, gq);
                WGRAD<KT, KT, 0, 0, LD, LD>(wg_q, wg_o, acc_w1q);
                float back[KT][4];
                chain_t<KT, KT, LD>(w1q + lt_d, gq, back);   // (by rows of the one image, with or without TR)
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gq[t][r] = back[t][r];
            }
            if (mlp_q)
                metanet_bwd(gq, zhq, rstd_q, lnq_g, agq, abq, 0, hq, q0, w2qT, w1qT, w1q, acc_w1q, acc_w2q);
            if (mlp_k) {
                if constexpr (SHLN)   // one table and one LayerNorm: both roles add into the same accumulators
                    metanet_bwd(gk, zhk, rstd_k, lnk_g, agq, abq, 8, hk, k0, w2kT, w1kT, w1k, acc_w1q, acc_w2q);
                else if constexpr (SAME)
                    metanet_bwd(gk, zhk, rstd_k, lnk_g, agk, abk, 8, hk, k0, w2kT, w1kT, w1k, acc_w1q, acc_w2q);
                else
                    metanet_bwd(gk, zhk, rstd_k, lnk_g, agk, abk, 8, hk, k0, w2kT, w1kT, w1k, acc_w1k, acc_w2k);
            }

            // projections: dW{q,k,v}[i][o] += x^T g ; dx = dr + gq Wq^T + gk Wk^T + gv Wv^T
            // (the same interleaving as in metanet_bwd: the back-projection of a gradient - registers and weights only - runs between
            //  the LDS stores of that gradient's rows and the weight-gradient product that reads them back)
            float gv[KT][4], back[KT][4];
            auto back_dd = [&](float* img, const float (&in_)[KT][4], float (&out_)[KT][4]) {
                if constexpr (TR) chain<KT, KT, LD>(img + lo_d, in_, out_);
                else chain_t<KT, KT, LD>(img + lt_d, in_, out_);
            };
            auto add_back = [&]() {
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dr[t][r] += back[t][r];
            };
#ifdef SATRANS_EXP_WGRAD_INORDER
            store_frag<KT>(my_q, x, valid);
            store_frag<KT>(my_o, gq);
            WGRAD<KT, KT, 0, 0, LD, LD>(wg_q, wg_o, acc_wq);
            store_frag<KT>(my_o, gk);
            WGRAD<KT, KT, 0, 0, LD, LD>(wg_q, wg_o, acc_wk);
            WGRAD<KT, KT, 0, 0, LD, LD>(wg_q, wg_v, acc_wv);
            load_frag<KT>(my_v, gv, valid);
            back_dd(wqT, gq, back); add_back();
            back_dd(wkT, gk, back); add_back();
            back_dd(wvT, gv, back); add_back();
#else
            // (dx is SUMMED in the order q, k, v of every earlier round - the value term is computed first and added last: the
            //  order of a sum is part of a build's bits, and with the value term first the default bench run ended in its other
            //  attractor, memorising instead of collapsing onto the base rate: tools/experiments/README.md, round 6)
            float back_v[KT][4];
            store_frag<KT>(my_q, x, valid);
            store_frag<KT>(my_o, gq);
            load_frag<KT>(my_v, gv, valid);                                    // (the value gradients: rows phase E left in LDS)
            back_dd(wvT, gv, back_v);
            WGRAD<KT, KT, 0, 0, LD, LD>(wg_q, wg_o, acc_wq);
            WGRAD<KT, KT, 0, 0, LD, LD>(wg_q, wg_v, acc_wv);
            store_frag<KT>(my_o, gk);
            back_dd(wqT, gq, back); add_back();
            WGRAD<KT, KT, 0, 0, LD, LD>(wg_q, wg_o, acc_wk);
            back_dd(wkT, gk, back); add_back();
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dr[t][r] += back_v[t][r];
#endif
            if (valid) store_frag<KT>(dx + ((size_t)b * F + f) * D + g4, dr);
        }
        lds_barrier();
        STAMP(6);
      }
      // ---- this scenario's generated-weight gradients: record (workgroup + scenario) ---------------------------------
      // with one shared table both roles accumulated into acc_w1q / acc_w2q; the reducer reads the part of a role only
      // when that role is active, so the sums go to the Q part when Q is modulated, else to the K part
      if (gate) {
          // gate vectors: [dvec_q D | dvec_k D] at the head of the record; per-lane sums -> token lanes -> waves, fixed order
          float* rec = records + (size_t)(blockIdx.x + scen) * TSZ;
          auto flush_vec = [&](float (&ag)[KT][4], float* dst) {
#pragma unroll
              for (int t = 0; t < KT; ++t)
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                      float sv_ = ag[t][r];
#pragma unroll
                      for (int m = 1; m < 16; m <<= 1) sv_ += __shfl_xor(sv_, m, 64);
                      if (n == 0) stage[wave * D + 16 * t + g4 + r] = sv_;
                      ag[t][r] = 0.f;
                  }
              __syncthreads();
              for (int e = threadIdx.x; e < D; e += kFusedBlock)
                  dst[e] = 2.0f * (((stage[e] + stage[D + e]) + stage[2 * D + e]) + stage[3 * D + e]);
              __syncthreads();
          };
          flush_vec(agq, rec);
          flush_vec(agk, rec + D);
      } else if (bilin) {
          float* rec = records + (size_t)(blockIdx.x + scen) * TSZ;      // the full D x D product; the reducer keeps the H blocks
          f32x4 tmp[KT][KT];
#pragma unroll
          for (int i = 0; i < KT; ++i)
#pragma unroll
              for (int j = 0; j < KT; ++j) { tmp[i][j] = acc_w1q[i][j]; acc_w1q[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
          flush(tmp, KTc{}, KTc{}, rec, true);
      } else {
          float* rec = records + (size_t)(blockIdx.x + scen) * TSZ;
          const bool to_k = SAME && !meta_q;
          flush(acc_w1q, KTc{}, UTc{}, rec + (to_k ? 2 * D * U : 0), true);
          flush(acc_w2q, UTc{}, KTc{}, rec + (to_k ? 3 * D * U : D * U), true);
          flush(acc_w1k, KTc{}, UTc{}, rec + (to_k ? 0 : 2 * D * U), !SAME);
          flush(acc_w2k, UTc{}, KTc{}, rec + (to_k ? D * U : 3 * D * U), !SAME);
      }
    }

    STAMP(7);
    // ---- scenario-independent gradients of this workgroup ------------------------------------------------------------------
    flush(acc_wq, KTc{}, KTc{}, common, true);
    flush(acc_wk, KTc{}, KTc{}, common + D * D, true);
    flush(acc_wv, KTc{}, KTc{}, common + 2 * D * D, true);
    flush(acc_wo, KTc{}, KTc{}, common + 3 * D * D, true);
    auto flush_ln = [&](float (&ag)[KT][4], float (&ab)[KT][4], int off) {
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sg_ = ag[t][r], sb_ = ab[t][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    sg_ += __shfl_xor(sg_, m, 64);
                    sb_ += __shfl_xor(sb_, m, 64);
                }
                if (n == 0) {
                    stage[wave * 2 * D + 16 * t + g4 + r] = sg_;
                    stage[wave * 2 * D + D + 16 * t + g4 + r] = sb_;
                }
            }
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * D; e += kFusedBlock)
            common[off + e] = ((stage[e] + stage[2 * D + e]) + stage[4 * D + e]) + stage[6 * D + e];
        __syncthreads();
    };
    flush_ln(agl, abl, 4 * D * D);
    if constexpr (SHLN) {
        // one MetaNet LayerNorm for both roles: the sums of both sit in agq / abq.  The reducer adds the Q and the K slot when
        // both roles are modulated and reads the K slot alone when only K is (fused_common_reduce)
        const int off_sum = 4 * D * D + (mlp_q ? 2 * D : 4 * D), off_zero = 4 * D * D + (mlp_q ? 4 * D : 2 * D);
        flush_ln(agq, abq, off_sum);
        for (int e = threadIdx.x; e < 2 * D; e += kFusedBlock) common[off_zero + e] = 0.f;
    } else {
        flush_ln(agq, abq, 4 * D * D + 2 * D);
        flush_ln(agk, abk, 4 * D * D + 4 * D);
    }
    if constexpr (HEADF) {
        // ---- the head's gradients and the loss of this workgroup's samples: one row [g_w (F D + n_dense) | g_b | loss] per
        //      workgroup, added up in workgroup order by head_reduce_kernel (fixed order: bitwise reproducible) ------------
        const int ncol = F * D + hd.n_dense;
        float* prow = hd.partial + (size_t)blockIdx.x * (ncol + 2);
        if (idle) {
            for (int c = threadIdx.x; c < ncol + 2; c += kFusedBlock) prow[c] = 0.f;
        } else {
            const int tok_ = 16 * wave + n, ls_ = tok_ / F;
            if (g == 0 && tok_ - ls_ * F == 0 && ls_ < Tsamp) { s_fin[ls_] = head_loss; s_fin[64 + ls_] = head_gb; }
            __syncthreads();
            for (int e = threadIdx.x; e < F * D; e += kFusedBlock) {
                const int f_ = e / D, feat = e - f_ * D;
                const int t_ = feat >> 4, g_ = (feat & 15) >> 2, r_ = feat & 3;
                float acc = 0.f;
                for (int ls = 0; ls < Tsamp; ++ls) {             // the sample slots of a tile, in slot order
                    const int tk = ls * F + f_;
                    if (tk < 64) acc += s_hw[(size_t)((tk >> 4) * 64 + g_ * 16 + (tk & 15)) * 4 * KT + 4 * t_ + r_];
                }
                prow[e] = acc;
            }
            if ((int)threadIdx.x < hd.n_dense) {
                float acc = 0.f;
                for (int ls = 0; ls < Tsamp; ++ls) acc += s_hwd[ls * kHeadDenseMax + threadIdx.x];
                prow[F * D + threadIdx.x] = acc;
            }
            if (threadIdx.x == 0) {
                float gb = 0.f, lsum = 0.f;
                for (int ls = 0; ls < Tsamp; ++ls) { lsum += s_fin[ls]; gb += s_fin[64 + ls]; }
                prow[ncol] = gb;
                prow[ncol + 1] = lsum;
            }
        }
    }
}


// -------------------------------------------------------------------------------------------------------------------
#undef WGRAD
// Reduction of the backward kernel's output, fixed order (bitwise reproducible):
//   common  [G][CSZ]   CSZ = 4*D*D + 6*D : [wq|wk|wv|wo|ln g,b|lnq g,b|lnk g,b] per workgroup
//   records [G+S][TSZ] TSZ = 4*D*U       : [w1q|w2q|w1k|w2k] of (workgroup w, scenario s) at index w + s
// -------------------------------------------------------------------------------------------------------------------
// One launch per layer.  Blocks [0, common_blocks): the scenario-independent part; the others: the generated-weight records.
// Both use blocks of 32 elements x kRG = 32 groups of workgroups: every group adds its contiguous share of the workgroup range in
// index order, the group sums are then combined in group order (fixed order => bitwise reproducible).  (8 groups: 32 dependent
// rounds of loads per thread, 12 us per launch for 4 MB; 32 groups: 8 rounds.)
constexpr int kRG = kReduceGroups;      // (common.h)
constexpr int kReduceMaxLayers = 8;      // layers one reduction launch can serve (satrans_layer_bwd_reduce)
__device__ __forceinline__ void fused_common_reduce(const float* __restrict__ common, int G, int D, int flags, int block,
                                                    float* g_wq, float* g_wk, float* g_wv, float* g_wo, float* g_ln,
                                                    float* g_lnq, float* g_lnk) {
    __shared__ float s_a[kRG][32], s_b[kRG][32];
    const int CSZ = 4 * D * D + 6 * D, DD = D * D;
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int e = block * 32 + lane;
    const bool plain = !(flags & (SATRANS_GATE | SATRANS_BILINEAR));        // gate / bilinear: no MetaNet, no MetaNet LayerNorm
    const bool mq = plain && (flags & SATRANS_META_Q) && g_lnq, mk = plain && (flags & SATRANS_META_K) && g_lnk;
    // without 'pos' the Q and K MetaNets share ONE LayerNorm (satrans.py:46): one thread adds both roles, Q first
    const bool shared = mq && mk && g_lnq == g_lnk;
    const int r = e - 4 * DD;
    const bool second = shared && r >= 2 * D && r < 4 * D;       // this thread also needs the K-role element e + 2D
    const int share = (G + kRG - 1) / kRG;
    const int lo = grp * share, hi = min(G, lo + share);
    float acc = 0.f, acc2 = 0.f;
    if (e < CSZ) {
#pragma unroll 4
        for (int w = lo; w < hi; ++w) {
            acc += common[(size_t)w * CSZ + e];
            if (second) acc2 += common[(size_t)w * CSZ + e + 2 * D];
        }
    }
    s_a[grp][lane] = acc;
    s_b[grp][lane] = acc2;
    __syncthreads();
    if (grp != 0 || e >= CSZ) return;
    float t = 0.f, t2 = 0.f;
    for (int k = 0; k < kRG; ++k) { t += s_a[k][lane]; t2 += s_b[k][lane]; }
    if (e < DD) { g_wq[e] += t; return; }
    if (e < 2 * DD) { g_wk[e - DD] += t; return; }
    if (e < 3 * DD) { g_wv[e - 2 * DD] += t; return; }
    if (e < 4 * DD) { g_wo[e - 3 * DD] += t; return; }
    if (r < 2 * D) { g_ln[r] += t; return; }                     // [ln 2D | lnq 2D | lnk 2D]
    if (r < 4 * D) {
        if (shared) g_lnq[r - 2 * D] += t + t2;
        else if (mq) g_lnq[r - 2 * D] += t;
        return;
    }
    if (mk && !shared) g_lnk[r - 4 * D] += t;
}

// `n_layers` record sets (one per layer of a step, in the order their backward kernels ran) whose gradients go to generated-weight
// tables that MAY BE THE SAME MEMORY for every layer (no 'pos' flag: one table for all layers): the block adds them one layer after
// the other, which is also the order - and hence the bits - of one reduction launch per layer.
__device__ __forceinline__ void fused_records_reduce(int n_layers, const float* const* records_of, float* const* g_tab_q_of,
                                                     float* const* g_tab_k_of, const int32_t* __restrict__ seg,
                                                     int S, int T, int G, int D, int U, int H, int flags, int64_t tab_stride,
                                                     int block, int s) {
    __shared__ float s_q[kRG][32], s_k[kRG][32];
    // One record per (workgroup, scenario), TSZ = 4 D U floats apart.  What it holds and which elements of the generated row
    // they are:   MetaNet   [W1q | W2q | W1k | W2k]: role r = elements [r * 2DU, (r + 1) * 2DU) = the row itself
    //             gate      [dvec_q D | dvec_k D]
    //             bilinear  the full D x D product q0^T gq; row element (h, i, j) = entry (h d + i, h d + j)   (queries only)
    const int TSZ = 4 * D * U, d = D / H;
    const bool gate = flags & SATRANS_GATE, bil = flags & SATRANS_BILINEAR;
    const int half = gate ? D : (bil ? D * d : 2 * D * U);      // elements of one role's generated row
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int e = block * 32 + lane;
    // the workgroups whose tile range intersects scenario s (same arithmetic as work_range in the kernel)
    int total = 0, pre = 0;
    for (int k = 0; k < S; ++k) {
        if (k == s) pre = total;
        total += tiles_of(seg, k, T);
    }
    const int nt = tiles_of(seg, s, T);
    if (nt == 0) return;                                        // (uniform over the block)
    const int per = (total + G - 1) / G;
    const int w_lo = pre / per, w_hi = (pre + nt - 1) / per;
    const int share = (w_hi - w_lo + kRG) / kRG;
    const int a = w_lo + grp * share, b = min(w_hi + 1, a + share);
    const bool mq = bil || (flags & SATRANS_META_Q), mk = !bil && (flags & SATRANS_META_K);
    int src_q = e, src_k = half + e;
    if (bil && e < half) {
        const int h = e / (d * d), i = (e / d) % d, j = e % d;
        src_q = (h * d + i) * D + h * d + j;
    }
    // the partial sums of ALL layers first (their loads are independent: one round of memory latency instead of one per layer), then
    // the group sums layer after layer through LDS
    float aq[kReduceMaxLayers], ak[kReduceMaxLayers];
#pragma unroll
    for (int l = 0; l < kReduceMaxLayers; ++l) {
        aq[l] = 0.f; ak[l] = 0.f;
        if (l < n_layers && e < half) {
            const float* records = records_of[l];
#pragma unroll 4
            for (int w = a; w < b; ++w) {
                const float* rec = records + (size_t)(w + s) * TSZ;
                if (mq) aq[l] += rec[src_q];
                if (mk) ak[l] += rec[src_k];
            }
        }
    }
#pragma unroll
    for (int l = 0; l < kReduceMaxLayers; ++l) {
        if (l >= n_layers) break;
        s_q[grp][lane] = aq[l];
        s_k[grp][lane] = ak[l];
        __syncthreads();
        if (grp == 0 && e < half) {
            float tq = 0.f, tk = 0.f;
            for (int k = 0; k < kRG; ++k) { tq += s_q[k][lane]; tk += s_k[k][lane]; }
            float* g_tab_q = g_tab_q_of[l];
            float* g_tab_k = g_tab_k_of[l];
            if (mq && mk && g_tab_q == g_tab_k) {
                g_tab_q[(size_t)s * tab_stride + e] += tq + tk;
            } else {
                if (mq) g_tab_q[(size_t)s * tab_stride + e] += tq;
                if (mk) g_tab_k[(size_t)s * tab_stride + e] += tk;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(32 * kRG) void fused_reduce_kernel(const float* __restrict__ common, const float* __restrict__ records,
                                                         const int32_t* __restrict__ seg, int S, int T, int G, int D, int U,
                                                         int H, int flags, int64_t tab_stride, int common_blocks, int record_blocks,
                                                         float* g_wq, float* g_wk, float* g_wv, float* g_wo, float* g_ln,
                                                         float* g_lnq, float* g_lnk, float* g_tab_q, float* g_tab_k) {
    const int bx = blockIdx.x;
    if (bx < common_blocks) {
        fused_common_reduce(common, G, D, flags, bx, g_wq, g_wk, g_wv, g_wo, g_ln, g_lnq, g_lnk);
    } else {
        const int rb = bx - common_blocks;
        const float* recs[1] = {records};
        float *gq[1] = {g_tab_q}, *gk[1] = {g_tab_k};
        fused_records_reduce(1, recs, gq, gk, seg, S, T, G, D, U, H, flags, tab_stride, rb % record_blocks, rb / record_blocks);
    }
}

// ---- all layers of a step in ONE reduction launch (satrans_layer_bwd_reduce): the backward kernels of the step's layers were
//      launched back to back with a slab buffer each; blocks [l * per_layer, (l + 1) * per_layer) do what fused_reduce_kernel does
//      for layer l, the blocks behind them add the fused head's partial rows (head_reduce_kernel's arithmetic and order) ----------
struct ReduceLayers {
    int n;
    const float* slabs[kReduceMaxLayers];
    float* g[kReduceMaxLayers][9];      // g_wq, g_wk, g_wv, g_wo, g_ln, g_lnq, g_lnk, g_tab_q, g_tab_k
};
struct ReduceHead {
    const float* partial;               // [nblk][ncol + 2] or null
    int nblk, ncol;
    float *g_w, *g_b;
    double* loss_sum;
};
__device__ __forceinline__ void head_partials_reduce(const ReduceHead& hd, int block) {
    __shared__ double s_acc[kRG][32];
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = block * 32 + lane, ncol = hd.ncol;
    const int share = (hd.nblk + kRG - 1) / kRG;
    const int lo = grp * share, hi = min(hd.nblk, lo + share);
    double acc = 0.0;
    if (c < ncol + 2) {
        if (c == ncol + 1) {
            for (int k = lo; k < hi; ++k) acc += (double)hd.partial[(size_t)k * (ncol + 2) + c];
        } else {
            float a32 = 0.f;
#pragma unroll 8
            for (int k = lo; k < hi; ++k) a32 += hd.partial[(size_t)k * (ncol + 2) + c];
            acc = (double)a32;
        }
    }
    s_acc[grp][lane] = acc;
    __syncthreads();
    if (grp != 0 || c >= ncol + 2) return;
    if (c == ncol + 1) {
        double t = 0.0;
        for (int k = 0; k < kRG; ++k) t += s_acc[k][lane];
        hd.loss_sum[0] += t;
        return;
    }
    float t = 0.f;
    for (int k = 0; k < kRG; ++k) t += (float)s_acc[k][lane];
    if (c < ncol) hd.g_w[c] += t;
    else hd.g_b[0] += t;
}
__global__ __launch_bounds__(32 * kRG) void fused_reduce_all_kernel(ReduceLayers R, ReduceHead hd, const int32_t* __restrict__ seg,
                                                             int S, int T, int G, int D, int U, int H, int flags,
                                                             int64_t tab_stride, int common_blocks, int record_blocks,
                                                             int per_layer) {
    __builtin_amdgcn_s_setprio(3);      // (ahead of the side stream's next-batch kernels when they share a SIMD)
    // blocks: [n x common_blocks: the scenario-independent part of every layer] [record_blocks x S: the generated-weight records of
    // ALL layers, layer after layer inside the block - their tables may be one and the same] [the fused head's partial rows]
    int bx = blockIdx.x;
    if (bx < R.n * common_blocks) {
        const int l = bx / common_blocks;
        float* const* g = R.g[l];
        fused_common_reduce(R.slabs[l], G, D, flags, bx - l * common_blocks, g[0], g[1], g[2], g[3], g[4], g[5], g[6]);
        return;
    }
    bx -= R.n * common_blocks;
    const int n_rec = per_layer - common_blocks;      // record_blocks * S, or 0 without generated weights
    if (bx < n_rec) {
        const float* recs[kReduceMaxLayers];
        float *gq[kReduceMaxLayers], *gk[kReduceMaxLayers];
        for (int l = 0; l < R.n; ++l) {
            recs[l] = R.slabs[l] + (size_t)G * (4 * D * D + 6 * D);
            gq[l] = R.g[l][7];
            gk[l] = R.g[l][8];
        }
        fused_records_reduce(R.n, recs, gq, gk, seg, S, T, G, D, U, H, flags, tab_stride, bx % record_blocks, bx / record_blocks);
        return;
    }
    head_partials_reduce(hd, bx - n_rec);
}

// -------------------------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------------------------

// The field count the fused backward is instantiated with for this layer (0: the runtime-F instantiation).  With a constant the
// backward runs the matrix-pipe attention arm and takes the hand-over straight into registers (REGH) - and the forward then
// writes the numerators as padded rows: ONE predicate for both launchers.
static int fused_bwd_ft(const satrans_layer_desc* d) {
    if (d->D != 32 || d->H != 4) return 0;
    const bool same = d->tab_q == d->tab_k && d->lnq_g == d->lnk_g && d->lnq_b == d->lnk_b;
    const int mod = (d->flags & SATRANS_GATE) ? 1 : ((d->flags & SATRANS_BILINEAR) ? 2 : 0);
    if (mod) return same && d->F == 19 ? 19 : 0;
    return same ? (d->F == 19 ? 19 : 0) : (d->F == 15 ? 15 : 0);
}
#ifdef SATRANS_HANDOVER_LDS
static bool fused_save_rows(const satrans_layer_desc* d) { return false; }
#else
static bool fused_save_rows(const satrans_layer_desc* d) { return kAttnMfma && fused_bwd_ft(d) != 0; }
#endif

static int64_t fused_fwd_lds_floats(int T, int F, int D, int U, bool same_tab) {
    const int LD = D + 4, LU = U + 4;
    const int64_t rows = (((int64_t)T * F + 3 + 15) / 16) * 16;      // (as the kernel: three rows of slack behind the last sample)
    const int64_t dd = (int64_t)D * LD;
    const int64_t mlp = (int64_t)D * LU + (int64_t)U * LD;
    return 4 * dd + (same_tab ? 1 : 2) * mlp + 6 * D + 4 * rows * LD + 64;
}

template <int D, int U, int H, int WAVES, int MOD = 0, bool SAVE = false, bool PROW = false>
static int launch_fwd_w(const satrans_layer_desc* d, float* y, float* att, hipStream_t stream) {
    if constexpr (SAVE && !PROW && D == 32)
        if (fused_save_rows(d)) return launch_fwd_w<D, U, H, WAVES, MOD, SAVE, true>(d, y, att, stream);
    const bool same_tab = d->tab_q == d->tab_k;
    // samples per tile: as many as keep `per_cu` workgroups per CU, preferring tiles that fill their 16-token MFMA rows
    int best = 0;
    double best_eff = 0.0;
    const int budgets[2] = {WAVES > 4 ? 156 * 1024 : 78 * 1024, 156 * 1024};   // two 4-wave workgroups per CU if any tile fits
    for (int budget : budgets) {
        for (int t = 1; t <= 4 * WAVES; ++t) {
            if (fused_fwd_lds_floats(t, d->F, D, U, same_tab) * 4 > budget) break;
            const int tok = t * d->F, ntt = (tok + 15) / 16;
            const double eff = (double)tok / (16.0 * ntt) * (double)ntt / (double)(ceil_div(ntt, WAVES) * WAVES);
            if (eff >= best_eff) { best_eff = eff; best = t; }
        }
        if (best) break;
    }
    SATRANS_REQUIRE(best > 0, SATRANS_E_UNSUPPORTED, "layer_fwd(fused): F=%d does not fit LDS", d->F);
    const size_t lds = (size_t)fused_fwd_lds_floats(best, d->F, D, U, same_tab) * 4;
    static size_t attr_set = 0;
    if (lds > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)layer_fwd_fused_kernel<D, U, H, WAVES, MOD, SAVE, PROW>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "layer_fwd(fused): LDS attribute: %s", hipGetErrorString(e));
        attr_set = lds;
    }
    const int64_t tiles = ceil_div(d->B, best);                        // (the kernel splits the batch by samples)
    const int per_cu = lds * 2 <= (size_t)160 * 1024 ? 2 : 1;
    const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, (int64_t)cu_count() * per_cu));
    if (KernelTimer::Pair* tp = g_ktimer.next(0))      // (satrans_kernel_timing: the dispatch's own begin / end timestamps)
        hipExtLaunchKernelGGL((layer_fwd_fused_kernel<D, U, H, WAVES, MOD, SAVE, PROW>), dim3(gx), dim3(64 * WAVES), lds, stream,
                              tp->start, tp->stop, 0, *d, best, y, att);
    else
    layer_fwd_fused_kernel<D, U, H, WAVES, MOD, SAVE, PROW><<<gx, 64 * WAVES, lds, stream>>>(*d, best, y, att);
    SATRANS_CHECK_LAUNCH("layer_fwd_fused_kernel");
    return SATRANS_OK;
}

template <int D, int U, int H>
static int launch_fwd(const satrans_layer_desc* d, float* y, float* att, hipStream_t stream) {
    return launch_fwd_w<D, U, H, kFusedWaves>(d, y, att, stream);
}

}  // namespace satrans

extern "C" int satrans_layer_fused_supported(const satrans_layer_desc* d);
extern "C" int64_t satrans_layer_attn_save_floats_fused(const satrans_layer_desc* d);

namespace satrans {

static int64_t fused_bwd_lds_floats(int T, int F, int D, int U, int H, bool same_tab, bool tr, bool head = false) {
    const int LD = D + 4, LU = U + 4;
    auto r4 = [](int64_t v) { return (v + 3) & ~(int64_t)3; };
    const int64_t tasks = (int64_t)T * H * F;
    const int copies = tr ? 2 : 1;         // forward images, and their transposes when asked for
    const int64_t dd = (int64_t)D * LD;
    const int64_t mlp = (int64_t)D * LU + (int64_t)U * LD;
    // (fused head: head weights, token dots, dense terms, final sums, per-lane and per-slot gradient sums - the kernel's take() calls)
    const int64_t hd = head ? r4((int64_t)F * D + kHeadDenseMax) + 64 + 64 + 128 + (int64_t)kFusedBlock * 4 * (D / 16) + 64 * kHeadDenseMax : 0;
    // (the dS cache doubles as the staging area of the saved-attention hand-over: dy rows + attention outputs of a tile)
    const int64_t FP = (F + 3) & ~3;       // (the matrix-pipe arm of the attention backward keeps dS padded: [T H][FP][FP])
    const int64_t ds = std::max<int64_t>(std::max<int64_t>(tasks * F, 2 * (int64_t)T * F * D), (int64_t)T * H * FP * FP);
    return copies * 4 * dd + (same_tab ? 1 : 2) * copies * mlp + 6 * D + 5 * 64 * LD + 2 * r4(tasks) + r4(tasks * F) + r4(ds) + hd + 64;
}

// MetaNet width the fused kernels are INSTANTIATED with for an embedding dim (U = 2 D; forward-only D = 64: 16).  With a MetaNet
// d->U equals it (satrans_layer_fused_supported); gate / bilinear layers ignore d->U, but the kernels' LDS layout and their
// generated-row records are laid out with the template width whatever d->U says - every size below must use this one.
static int fused_width(const satrans_layer_desc* d) { return d->D == 64 ? 16 : 2 * d->D; }

struct FusedBwdPlan {
    int T, G;
    size_t lds;
    bool tr;      // transposed weight images in LDS
};

static bool fused_bwd_plan(const satrans_layer_desc* d, FusedBwdPlan& p, bool head = false) {
    if (!satrans_layer_fused_supported(d) || d->D > 32 || d->F > 32) return false;   // keep bits: one 32-bit word per row
    const bool same_tab = d->tab_q == d->tab_k && d->lnq_g == d->lnk_g && d->lnq_b == d->lnk_b;   // (as satrans_layer_bwd_fused picks SAME)
    p.T = 64 / d->F;
    // No transposed copies of the weight images: the backward products read the forward images by rows (chain_t).  The
    // four-way bank conflicts of those reads cost nothing measurable, while staging half as many images per workgroup
    // makes the kernel 4 % faster (0.934 -> 0.895 ms per step) and leaves 37 KB of LDS free.
    p.tr = false;      // (round 1's transposed copies: measured slower, the instantiation is no longer built)
    const int Uw = fused_width(d);
    p.lds = (size_t)fused_bwd_lds_floats(p.T, d->F, d->D, Uw, d->H, same_tab, p.tr, head) * 4;
    if (p.lds > 160 * 1024) return false;
    const int64_t tiles = ceil_div(d->B, p.T) + d->S;
    p.G = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, cu_count()));      // one workgroup per CU, one round
    return true;
}

template <int D, int U, int H, bool SAME, bool TR, int FT = 0, int MOD = 0, bool SAVE = false, bool HEADF = false>
static int launch_bwd(const satrans_layer_desc* d, const FusedBwdPlan& p, const float* dy, float* dx, float* slabs,
                      hipStream_t stream, const FusedHeadArgs* hd = nullptr) {
    static size_t attr_set = 0;
    if (p.lds > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)layer_bwd_fused_kernel<D, U, H, SAME, TR, FT, MOD, SAVE, HEADF>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "layer_bwd(fused): LDS attribute: %s", hipGetErrorString(e));
        attr_set = p.lds;
    }
    if (KernelTimer::Pair* tp = g_ktimer.next(HEADF ? 2 : 1))
        hipExtLaunchKernelGGL((layer_bwd_fused_kernel<D, U, H, SAME, TR, FT, MOD, SAVE, HEADF>), dim3(p.G), dim3(kFusedBlock),
                              p.lds, stream, tp->start, tp->stop, 0, *d, p.T, dy, dx, slabs, hd ? *hd : FusedHeadArgs{});
    else
    layer_bwd_fused_kernel<D, U, H, SAME, TR, FT, MOD, SAVE, HEADF><<<p.G, kFusedBlock, p.lds, stream>>>(
        *d, p.T, dy, dx, slabs, hd ? *hd : FusedHeadArgs{});
    SATRANS_CHECK_LAUNCH("layer_bwd_fused_kernel");
    return SATRANS_OK;
}

}  // namespace satrans

using namespace satrans;

// 1 when the fused kernels are built for this shape
extern "C" int satrans_layer_fused_supported(const satrans_layer_desc* d) {
    if (!d) return 0;
    if ((d->flags & SATRANS_GATE) && (d->flags & SATRANS_BILINEAR)) return 0;
    if (d->F > 64) return 0;
    const int D = d->D, U = d->U, H = d->H;
    const bool alt = d->flags & (SATRANS_GATE | SATRANS_BILINEAR);      // gate / bilinear: no MetaNet, its width plays no part
    const bool meta = (d->flags & (SATRANS_META_Q | SATRANS_META_K)) && !alt;
    if (D == 32 && H == 4 && (!meta || U == 64)) return 1;
    if (D == 16 && H == 2 && (!meta || U == 32)) return 1;
    if (D == 64 && H == 4 && (!meta || U == 16) && !alt) return 1;
    return 0;
}

extern "C" int satrans_layer_fwd_fused(const satrans_layer_desc* d, float* y, float* att, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(satrans_layer_fused_supported(d), SATRANS_E_UNSUPPORTED, "layer_fwd(fused): shape not built");
    // D = 32: one 12-wave workgroup per CU (three waves per SIMD at <= 168 VGPRs, one copy of the weight images in LDS for
    // ten samples per tile) beats two 4-wave workgroups (two waves per SIMD) by 11 % (0.275 vs 0.309 ms per step)
    const int mod = (d->flags & SATRANS_GATE) ? 1 : ((d->flags & SATRANS_BILINEAR) ? 2 : 0);
    const bool save = d->attn_save && d->D == 32 && d->F <= 32 && satrans_layer_attn_save_floats_fused(d) > 0;   // (as the backward decides)
    if (mod && d->D == 32)
        return mod == 1 ? (save ? launch_fwd_w<32, 64, 4, 12, 1, true>(d, y, att, stream) : launch_fwd_w<32, 64, 4, 12, 1>(d, y, att, stream))
                        : (save ? launch_fwd_w<32, 64, 4, 12, 2, true>(d, y, att, stream) : launch_fwd_w<32, 64, 4, 12, 2>(d, y, att, stream));
    if (mod) return mod == 1 ? launch_fwd_w<16, 32, 2, kFusedWaves, 1>(d, y, att, stream) : launch_fwd_w<16, 32, 2, kFusedWaves, 2>(d, y, att, stream);
    if (d->D == 32) return save ? launch_fwd_w<32, 64, 4, 12, 0, true>(d, y, att, stream) : launch_fwd_w<32, 64, 4, 12>(d, y, att, stream);
    if (d->D == 16) return launch_fwd<16, 32, 2>(d, y, att, stream);
    return launch_fwd<64, 16, 4>(d, y, att, stream);
}


extern "C" int satrans_layer_bwd_fused_supported(const satrans_layer_desc* d) {
    FusedBwdPlan p;
    return d && fused_bwd_plan(d, p) ? 1 : 0;
}

extern "C" int64_t satrans_layer_attn_save_floats_fused(const satrans_layer_desc* d) {
    FusedBwdPlan p;
    // built for the (32, 64, 4) MetaNet shape, one generated-weight table for both roles or one each (flag 'pos')
    if (!d || !fused_bwd_plan(d, p)) return 0;
    // (gate / bilinear: the attention state alone - they have no MetaNet rows; their layout reserves the slots all the same)
    const bool alt = d->flags & (SATRANS_GATE | SATRANS_BILINEAR);
    if (d->D != 32 || (!alt && (d->U != 64 || !(d->flags & (SATRANS_META_Q | SATRANS_META_K))))) return 0;
    const int64_t HF = (int64_t)d->H * d->F;
    // the forward saves from its register-resident score row (F <= 32), the backward copies the numerators 16 bytes at a time
    // (and stages the tile's dy and saved output rows in the dS cache: fused_bwd_lds_floats sizes it for both uses)
    if (d->F > 32 || (d->F * HF) % 4 != 0) return 0;
    // numerators [B][F][HF] | 1 / sum [B][HF] | keep words [B][HF] | attention output [B][F][D] | normalised MetaNet rows
    // [B][F][role][D] | their 1 / std [B][F][role]
    // (numerators: [B][F][HF], or - register hand-over - padded rows [B][H][F][FP]: sized for the larger)
    const int64_t FPs = (d->F + 3) & ~3;
    return (int64_t)d->B * ((int64_t)d->H * d->F * FPs + 2 * HF + (int64_t)d->F * d->D + (int64_t)d->F * (2 * d->D + 2));
}

extern "C" int64_t satrans_layer_bwd_slab_floats_fused(const satrans_layer_desc* d) {
    FusedBwdPlan p;
    if (!d) return -1;
    int64_t n = -1;
    if (fused_bwd_plan(d, p)) {
        const int64_t CSZ = 4 * (int64_t)d->D * d->D + 6 * d->D, TSZ = 4 * (int64_t)d->D * fused_width(d);
        n = std::max<int64_t>(n, (int64_t)p.G * CSZ + (int64_t)(p.G + d->S) * TSZ);
    }
    return n;
}

static int launch_fused_reduce(const satrans_layer_desc* d, const FusedBwdPlan& p, float* slabs, float* g_wq, float* g_wk,
                               float* g_wv, float* g_wo, float* g_ln, float* g_lnq, float* g_lnk, float* g_tab_q, float* g_tab_k,
                               hipStream_t stream) {
    const int D = d->D, U = fused_width(d), CSZ = 4 * D * D + 6 * D;      // (record stride: the kernel's template width)
    float* records = slabs + (size_t)p.G * CSZ;
    const bool meta = d->flags & (SATRANS_META_Q | SATRANS_META_K | SATRANS_BILINEAR);
    const int row_elems = (d->flags & SATRANS_GATE) ? D : ((d->flags & SATRANS_BILINEAR) ? D * (D / d->H) : 2 * D * U);
    SATRANS_REQUIRE(!meta || g_tab_q, SATRANS_E_BADARG, "layer_bwd: null generated-row gradient");
    const int common_blocks = (int)ceil_div(CSZ, 32), record_blocks = (int)ceil_div(row_elems, 32);
    fused_reduce_kernel<<<(unsigned)(common_blocks + (meta ? record_blocks * d->S : 0)), 32 * kRG, 0, stream>>>(
        slabs, records, d->seg, d->S, p.T, p.G, D, U, d->H, d->flags, d->tab_stride, common_blocks, record_blocks, g_wq, g_wk, g_wv,
        g_wo, g_ln, g_lnq, g_lnk, g_tab_q, g_tab_k);
    SATRANS_CHECK_LAUNCH("fused_reduce_kernel");
    return SATRANS_OK;
}

// the backward kernel alone: its per-workgroup slabs wait in `slabs` for satrans_layer_bwd_fused's own reduction launch or for
// satrans_layer_bwd_reduce_fused (one launch for all layers of a step)
extern "C" int satrans_layer_bwd_launch_fused(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FusedBwdPlan p;
    SATRANS_REQUIRE(dy && dx && slabs, SATRANS_E_BADARG, "layer_bwd: null pointer");
    // (a caller with one table but two LayerNorm vectors gets the two-table instantiation: it stages the table twice, nothing else)
    const bool same = d->tab_q == d->tab_k && d->lnq_g == d->lnk_g && d->lnq_b == d->lnk_b;
    int rc;
    {
    SATRANS_REQUIRE(fused_bwd_plan(d, p), SATRANS_E_UNSUPPORTED, "layer_bwd(fused): shape not built");
    SATRANS_REQUIRE((int64_t)d->B * d->F < ((int64_t)1 << 31), SATRANS_E_UNSUPPORTED, "layer_bwd(fused): B * F must stay below 2^31");
    constexpr bool f_const = true;
    // the AliCCP field count as a compile-time constant: -11 % (0.868 -> 0.777 ms over three layers).  The same for the 16
    // fields of the Alimama `sota-pos` shape (separate Q / K tables) spills 34 VGPRs and gains nothing: not instantiated.
    const int mod = (d->flags & SATRANS_GATE) ? 1 : ((d->flags & SATRANS_BILINEAR) ? 2 : 0);
    const bool save = d->D == 32 && d->attn_save && satrans_layer_attn_save_floats_fused(d) > 0;
    // (gate / bilinear at the AliCCP field count: the field count a constant, hence the matrix-pipe attention arm - round 6)
    const int ft = fused_bwd_ft(d);      // (the predicate the forward's hand-over layout follows: fused_save_rows)
    if (mod && save && ft == 19)
        rc = mod == 1 ? launch_bwd<32, 64, 4, true, false, 19, 1, true>(d, p, dy, dx, slabs, stream)
                      : launch_bwd<32, 64, 4, true, false, 19, 2, true>(d, p, dy, dx, slabs, stream);
    else if (mod && save)
        rc = mod == 1 ? (same ? launch_bwd<32, 64, 4, true, false, 0, 1, true>(d, p, dy, dx, slabs, stream)
                              : launch_bwd<32, 64, 4, false, false, 0, 1, true>(d, p, dy, dx, slabs, stream))
                      : (same ? launch_bwd<32, 64, 4, true, false, 0, 2, true>(d, p, dy, dx, slabs, stream)
                              : launch_bwd<32, 64, 4, false, false, 0, 2, true>(d, p, dy, dx, slabs, stream));
    else if (mod == 1)
        rc = d->D == 32 ? (same ? launch_bwd<32, 64, 4, true, false, 0, 1>(d, p, dy, dx, slabs, stream)
                                : launch_bwd<32, 64, 4, false, false, 0, 1>(d, p, dy, dx, slabs, stream))
                        : (same ? launch_bwd<16, 32, 2, true, false, 0, 1>(d, p, dy, dx, slabs, stream)
                                : launch_bwd<16, 32, 2, false, false, 0, 1>(d, p, dy, dx, slabs, stream));
    else if (mod == 2)
        rc = d->D == 32 ? (same ? launch_bwd<32, 64, 4, true, false, 0, 2>(d, p, dy, dx, slabs, stream)
                                : launch_bwd<32, 64, 4, false, false, 0, 2>(d, p, dy, dx, slabs, stream))
                        : (same ? launch_bwd<16, 32, 2, true, false, 0, 2>(d, p, dy, dx, slabs, stream)
                                : launch_bwd<16, 32, 2, false, false, 0, 2>(d, p, dy, dx, slabs, stream));
    else if (save)
        rc = !same ? (ft == 15 ? launch_bwd<32, 64, 4, false, false, 15, 0, true>(d, p, dy, dx, slabs, stream)
                               : launch_bwd<32, 64, 4, false, false, 0, 0, true>(d, p, dy, dx, slabs, stream))      // flag 'pos': one table per role
             : ft == 19 ? launch_bwd<32, 64, 4, true, false, 19, 0, true>(d, p, dy, dx, slabs, stream)
                        : launch_bwd<32, 64, 4, true, false, 0, 0, true>(d, p, dy, dx, slabs, stream);
    else if (d->D == 32 && same && d->F == 19 && f_const)
        rc = launch_bwd<32, 64, 4, true, false, 19>(d, p, dy, dx, slabs, stream);
    else if (d->D == 32) rc = same ? launch_bwd<32, 64, 4, true, false>(d, p, dy, dx, slabs, stream)
                                   : launch_bwd<32, 64, 4, false, false>(d, p, dy, dx, slabs, stream);
    else rc = same ? launch_bwd<16, 32, 2, true, false>(d, p, dy, dx, slabs, stream)
                   : launch_bwd<16, 32, 2, false, false>(d, p, dy, dx, slabs, stream);
    }
    return rc;
}

extern "C" int satrans_layer_bwd_fused(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, float* g_wq,
                                       float* g_wk, float* g_wv, float* g_wo, float* g_ln, float* g_lnq, float* g_lnk,
                                       float* g_tab_q, float* g_tab_k, void* stream_) {
    SATRANS_REQUIRE(g_wq && g_wk && g_wv && g_wo && g_ln, SATRANS_E_BADARG, "layer_bwd: null pointer");
    int rc = satrans_layer_bwd_launch_fused(d, dy, dx, slabs, stream_);
    if (rc) return rc;
    FusedBwdPlan p;
    fused_bwd_plan(d, p);
    return launch_fused_reduce(d, p, slabs, g_wq, g_wk, g_wv, g_wo, g_ln, g_lnq, g_lnk, g_tab_q, g_tab_k, (hipStream_t)stream_);
}

// ---- last layer of a training step with the head fused in (include/satrans_hip.h: satrans_layer_bwd_head) ---------------------
extern "C" int satrans_head_reduce_partials(const float* partial, int nblk, int ncol, float* g_w, float* g_b, double* loss_sum,
                                            void* stream);

extern "C" int satrans_layer_bwd_head_fused_supported(const satrans_layer_desc* d, const satrans_head_desc* h) {
    FusedBwdPlan p;
    if (!d || !h) return 0;
    if (!(d->D == 32 || d->D == 16)) return 0;
    if (h->n_dense < 0 || h->n_dense > kHeadDenseMax) return 0;
    return fused_bwd_plan(d, p, true) ? 1 : 0;
}

extern "C" int64_t satrans_layer_bwd_head_scratch_floats_fused(const satrans_layer_desc* d, int n_dense) {
    FusedBwdPlan p;
    if (!d || !fused_bwd_plan(d, p, true)) return -1;
    return (int64_t)p.G * ((int64_t)d->F * d->D + n_dense + 2);
}

extern "C" int satrans_layer_bwd_head_launch_fused(const satrans_layer_desc* d, const satrans_head_desc* h, float* dx, float* slabs,
                                                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FusedBwdPlan p;
    SATRANS_REQUIRE(dx && slabs, SATRANS_E_BADARG, "layer_bwd_head: null pointer");
    SATRANS_REQUIRE(h->w && h->bias && h->labels && h->prob && h->scratch, SATRANS_E_BADARG, "layer_bwd_head: null head pointer");
    SATRANS_REQUIRE(h->n_dense == 0 || (h->dense && h->h_dense_cols), SATRANS_E_BADARG, "layer_bwd_head: dense columns without a matrix");
    SATRANS_REQUIRE(h->loss_kind >= SATRANS_LOSS_BCE && h->loss_kind <= SATRANS_LOSS_MAE, SATRANS_E_BADARG, "layer_bwd_head: loss kind %d", h->loss_kind);
    SATRANS_REQUIRE(satrans_layer_bwd_head_fused_supported(d, h) && fused_bwd_plan(d, p, true), SATRANS_E_UNSUPPORTED,
                    "layer_bwd_head: shape not built");
    SATRANS_REQUIRE((int64_t)d->B * d->F < ((int64_t)1 << 31), SATRANS_E_UNSUPPORTED, "layer_bwd_head: B * F must stay below 2^31");
    FusedHeadArgs hd;
    hd.w = h->w; hd.bias = h->bias; hd.labels = h->labels;
    hd.dense = h->n_dense ? h->dense : h->labels;               // (n_dense == 0: element 0 of a readable array, multiplied by nothing)
    hd.dense_stride = h->n_dense ? h->dense_stride : 0;
    for (int j = 0; j < kHeadDenseMax; ++j) hd.dense_col[j] = h->n_dense ? h->h_dense_cols[std::min(j, h->n_dense - 1)] : 0;
    hd.n_dense = h->n_dense; hd.loss_kind = h->loss_kind;
    hd.prob = h->prob; hd.logit = h->logit; hd.partial = h->scratch;
    const bool same = d->tab_q == d->tab_k && d->lnq_g == d->lnk_g && d->lnq_b == d->lnk_b;
    constexpr bool f_const = true;
    int rc;
    const int mod = (d->flags & SATRANS_GATE) ? 1 : ((d->flags & SATRANS_BILINEAR) ? 2 : 0);
    if (mod && d->D == 32 && same && d->F == 19 && f_const)
        rc = mod == 1 ? launch_bwd<32, 64, 4, true, false, 19, 1, false, true>(d, p, nullptr, dx, slabs, stream, &hd)
                      : launch_bwd<32, 64, 4, true, false, 19, 2, false, true>(d, p, nullptr, dx, slabs, stream, &hd);
    else if (mod == 1)
        rc = d->D == 32 ? (same ? launch_bwd<32, 64, 4, true, false, 0, 1, false, true>(d, p, nullptr, dx, slabs, stream, &hd)
                                : launch_bwd<32, 64, 4, false, false, 0, 1, false, true>(d, p, nullptr, dx, slabs, stream, &hd))
                        : (same ? launch_bwd<16, 32, 2, true, false, 0, 1, false, true>(d, p, nullptr, dx, slabs, stream, &hd)
                                : launch_bwd<16, 32, 2, false, false, 0, 1, false, true>(d, p, nullptr, dx, slabs, stream, &hd));
    else if (mod == 2)
        rc = d->D == 32 ? (same ? launch_bwd<32, 64, 4, true, false, 0, 2, false, true>(d, p, nullptr, dx, slabs, stream, &hd)
                                : launch_bwd<32, 64, 4, false, false, 0, 2, false, true>(d, p, nullptr, dx, slabs, stream, &hd))
                        : (same ? launch_bwd<16, 32, 2, true, false, 0, 2, false, true>(d, p, nullptr, dx, slabs, stream, &hd)
                                : launch_bwd<16, 32, 2, false, false, 0, 2, false, true>(d, p, nullptr, dx, slabs, stream, &hd));
    else if (d->D == 32 && same && d->F == 19 && f_const)
        rc = launch_bwd<32, 64, 4, true, false, 19, 0, false, true>(d, p, nullptr, dx, slabs, stream, &hd);
    else if (d->D == 32)
        rc = same ? launch_bwd<32, 64, 4, true, false, 0, 0, false, true>(d, p, nullptr, dx, slabs, stream, &hd)
             : d->F == 15 ? launch_bwd<32, 64, 4, false, false, 15, 0, false, true>(d, p, nullptr, dx, slabs, stream, &hd)
                          : launch_bwd<32, 64, 4, false, false, 0, 0, false, true>(d, p, nullptr, dx, slabs, stream, &hd);
    else
        rc = same ? launch_bwd<16, 32, 2, true, false, 0, 0, false, true>(d, p, nullptr, dx, slabs, stream, &hd)
                  : launch_bwd<16, 32, 2, false, false, 0, 0, false, true>(d, p, nullptr, dx, slabs, stream, &hd);
    return rc;
}

extern "C" int satrans_layer_bwd_head_fused(const satrans_layer_desc* d, const satrans_head_desc* h, float* dx, float* slabs,
                                            float* g_wq, float* g_wk, float* g_wv, float* g_wo, float* g_ln, float* g_lnq,
                                            float* g_lnk, float* g_tab_q, float* g_tab_k, void* stream_) {
    SATRANS_REQUIRE(g_wq && g_wk && g_wv && g_wo && g_ln && h->loss_sum && h->g_w && h->g_b, SATRANS_E_BADARG, "layer_bwd_head: null pointer");
    int rc = satrans_layer_bwd_head_launch_fused(d, h, dx, slabs, stream_);
    if (rc) return rc;
    FusedBwdPlan p;
    fused_bwd_plan(d, p, true);
    rc = launch_fused_reduce(d, p, slabs, g_wq, g_wk, g_wv, g_wo, g_ln, g_lnq, g_lnk, g_tab_q, g_tab_k, (hipStream_t)stream_);
    if (rc) return rc;
    return satrans_head_reduce_partials(h->scratch, p.G, d->F * d->D + h->n_dense, h->g_w, h->g_b, h->loss_sum, stream_);
}

// One reduction launch for the backward kernels of n layers of ONE step (same batch, same bucketing, same shape and flags: the
// layers of a model) that were launched with satrans_layer_bwd_launch / _head_launch into a slab buffer each, plus - `head` given -
// the fused head's partial rows.  Same arithmetic and order per layer as the per-layer reduction: the same bits.
extern "C" int satrans_layer_bwd_reduce_fused(int n, const satrans_layer_desc* const* descs, float* const* slabs,
                                              const satrans_layer_grads* grads, const satrans_head_desc* head, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(n >= 1 && n <= kReduceMaxLayers && descs && slabs && grads, SATRANS_E_BADARG, "layer_bwd_reduce: 1..%d layers", kReduceMaxLayers);
    const satrans_layer_desc* d = descs[0];
    FusedBwdPlan p;
    SATRANS_REQUIRE(fused_bwd_plan(d, p), SATRANS_E_UNSUPPORTED, "layer_bwd_reduce: shape not built");
    const int D = d->D, U = fused_width(d), CSZ = 4 * D * D + 6 * D;
    const bool meta = d->flags & (SATRANS_META_Q | SATRANS_META_K | SATRANS_BILINEAR);
    const int row_elems = (d->flags & SATRANS_GATE) ? D : ((d->flags & SATRANS_BILINEAR) ? D * (D / d->H) : 2 * D * U);
    const int common_blocks = (int)ceil_div(CSZ, 32), record_blocks = (int)ceil_div(row_elems, 32);
    const int per_layer = common_blocks + (meta ? record_blocks * d->S : 0);
    ReduceLayers R;
    R.n = n;
    for (int l = 0; l < n; ++l) {
        const satrans_layer_desc* e = descs[l];
        FusedBwdPlan q;
        SATRANS_REQUIRE(e->B == d->B && e->F == d->F && e->D == d->D && e->U == d->U && e->H == d->H && e->S == d->S &&
                        e->flags == d->flags && e->seg == d->seg && e->tab_stride == d->tab_stride && fused_bwd_plan(e, q) &&
                        q.G == p.G && q.T == p.T, SATRANS_E_BADARG, "layer_bwd_reduce: layer %d differs from layer 0 in shape, flags or bucketing", l);
        const satrans_layer_grads& g = grads[l];
        SATRANS_REQUIRE(slabs[l] && g.g_wq && g.g_wk && g.g_wv && g.g_wo && g.g_ln && (!meta || g.g_tab_q), SATRANS_E_BADARG,
                        "layer_bwd_reduce: null pointer (layer %d)", l);
        R.slabs[l] = slabs[l];
        float* gp[9] = {g.g_wq, g.g_wk, g.g_wv, g.g_wo, g.g_ln, g.g_lnq, g.g_lnk, g.g_tab_q, g.g_tab_k};
        for (int k = 0; k < 9; ++k) R.g[l][k] = gp[k];
    }
    ReduceHead hd{};
    int head_blocks = 0;
    if (head) {
        SATRANS_REQUIRE(head->scratch && head->g_w && head->g_b && head->loss_sum, SATRANS_E_BADARG, "layer_bwd_reduce: null head pointer");
        FusedBwdPlan ph;
        SATRANS_REQUIRE(fused_bwd_plan(d, ph, true), SATRANS_E_UNSUPPORTED, "layer_bwd_reduce: fused head not built for this shape");
        hd.partial = head->scratch; hd.nblk = ph.G; hd.ncol = d->F * d->D + head->n_dense;
        hd.g_w = head->g_w; hd.g_b = head->g_b; hd.loss_sum = head->loss_sum;
        head_blocks = (int)ceil_div(hd.ncol + 2, 32);
    }
    fused_reduce_all_kernel<<<(unsigned)(n * common_blocks + (per_layer - common_blocks) + head_blocks), 32 * kRG, 0, stream>>>(
        R, hd, d->seg, d->S, p.T, p.G, D, U, d->H, d->flags, d->tab_stride, common_blocks, record_blocks, per_layer);
    SATRANS_CHECK_LAUNCH("fused_reduce_all_kernel");
    return SATRANS_OK;
}

#ifdef SATRANS_STAMPS
extern "C" int satrans_debug_read_stamps(unsigned long long* h_out, int reset) {
    if (hipMemcpyFromSymbol(h_out, HIP_SYMBOL(satrans::g_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(satrans::g_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" int satrans_kernel_timing(int on) {
    const int was = satrans::g_ktimer.armed ? 1 : 0;
    satrans::g_ktimer.armed = on != 0;
    return was;
}

extern "C" int satrans_kernel_timing_read(int* kinds, float* ms, int max) {
    using namespace satrans;
    int n = 0;
    for (int i = 0; i < g_ktimer.used; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(g_ktimer.pairs[i].stop) != hipSuccess ||
            hipEventElapsedTime(&t, g_ktimer.pairs[i].start, g_ktimer.pairs[i].stop) != hipSuccess) {
            set_error("kernel_timing_read: %s", hipGetErrorString(hipGetLastError()));
            g_ktimer.used = 0;
            return SATRANS_E_LAUNCH;
        }
        if (n < max) {
            if (kinds) kinds[n] = g_ktimer.pairs[i].kind;
            if (ms) ms[n] = t;
            ++n;
        }
    }
    g_ktimer.used = 0;
    return n;
}
