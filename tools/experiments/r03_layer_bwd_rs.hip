// Meta_Transformer_Layer backward, ROLE-SPLIT version: two waves per SIMD without splitting any accumulator.
//
// Reference: models/satrans.py:50-100 (layer), models/submodules.py:77-103 (MetaNet) + autograd.  Same mathematics, inputs,
// outputs, slab layout and reduction launch as layer_bwd_fused_kernel (layer_fused.hip); what differs is how a 64-token tile
// sits on the CU:
//
//   * one workgroup of EIGHT waves per CU.  Every 16-token tile is owned by a PAIR of waves that share a SIMD (waves w and
//     w + 4): the Q-role wave carries x -> q0 -> MetaNet_Q, the output block (Out_linear, residual, LayerNorm, dWo) and the
//     backward of those; the K-role wave carries x -> k0, v -> MetaNet_K and the backward of those.  The two roles of a layer
//     are independent between the attention phases, so the two waves of a SIMD run the token phases CONCURRENTLY on the same
//     matrix pipe - one wave's LDS waits, LayerNorm / dropout VALU work and operand stores sit under the other's MFMAs - and
//     each needs only ITS role's token state (x, in0, hidden, normalised rows: 48 registers instead of 80) and ITS role's
//     weight-gradient accumulators (96 instead of 128): 256 registers per wave, two waves per SIMD, no accumulator split over
//     waves, no operand exchange, no extra barrier (the 8-wave kernel of round 2 split every accumulator by output tile and
//     paid for it with two LDS operands per MFMA and 11 barriers per tile).  288 MFMAs per wave and tile on either side;
//   * the attention phases (one task = one (sample, head, row)) run on all 512 lanes: a task is a PAIR of adjacent lanes that
//     split the keys (phases B, D: partial max / sums / rows combined with one DPP swap each) or the two outputs (phase E: one
//     lane dk_j, its partner dv_j).  Half the instructions per lane and two waves per SIMD: a lone wave issues one VALU
//     instruction per 4 cycles, two waves per SIMD one per 2 (MI355X_MICROARCH.md, cycle constants);
//   * the K-role wave's operand scratch for its weight-gradient products lives in the softmax caches, which are dead by then;
//   * LayerNorm gamma / beta gradients are reduced over the 16 token lanes as they are produced and kept in 8 registers;
//   * weight images carry an XOR swizzle (16-byte chunk c of row k is stored at chunk c ^ [4 <= k mod 16 < 12]) that makes the
//     by-rows 16-byte reads of the transposed products (chain_t) conflict-free as well as the by-columns 4-byte reads of the
//     forward products, and the token-contraction products read rows FOUR apart per lane group: no bank conflicts on any
//     operand path (PMC of round 2: SQ_LDS_BANK_CONFLICT 62 % of the LDS-active cycles);
//   * fixed summation orders everywhere (no float atomics): bitwise reproducible.
//
// Phases per tile (|| = workgroup barrier):
//   A  token  Q: x -> q0 -> MetaNet -> q rows          K: x -> k0, v -> MetaNet -> k, v rows                          ||
//   B  pairs  attention forward: numerators -> cache, 1/sum, keep word, o rows                                        ||
//   C  token  Q: Out_linear + residual + LayerNorm forward / backward, dWo, go rows     (K: idle)                     ||
//   D  pairs  softmax backward by rows: dS (cache), dq rows                                                           ||
//   E  pairs  by columns: lane 0 dk_j, lane 1 dv_j                                                                    ||
//   F  token  Q: MetaNet_Q backward, dWq, dx part    K: MetaNet_K backward, dWk, dWv, dx part -> LDS                  ||
//      Q: dx = parts + dr
#include "layer_fused_common.h"

namespace satrans {

// diagnostic build (-DSATRANS_STAMPS): phase cycles of wave 0 (Q role, slots 0-15) and wave 4 (K role, slots 16-31)
#ifdef SATRANS_STAMPS
#define RSTAMP(slot)                                                                              \
    do {                                                                                          \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                             \
        if (lane == 0 && tw == 0) atomicAdd(&g_stamps[(slot) + 16 * role], now_ - st_prev);       \
        st_prev = now_;                                                                           \
    } while (0)
#else
#define RSTAMP(slot)
#endif

constexpr int kRsWaves = 8;
constexpr int kRsBlock = 64 * kRsWaves;
constexpr int kRsRows = 64;     // token rows of a workgroup tile

// Workgroup barrier for LDS hand-offs inside the tile loop.  __syncthreads() makes hipcc wait for EVERY outstanding memory
// operation (s_waitcnt vmcnt(0)) in front of s_barrier, which would stall the waves on the global prefetches that are meant to
// fly across the phases (next tile's sample index and input row, dy); only the LDS traffic has to be complete here.  Global
// memory is never used to pass data between waves inside the loop.
__device__ __forceinline__ void rs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CTRL>
__device__ __forceinline__ float rs_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// the value of the partner lane of an attention pair (lanes 2p, 2p + 1): quad_perm [1,0,3,2]
__device__ __forceinline__ float rs_swap(float v) { return rs_dpp<0xB1>(v); }
__device__ __forceinline__ uint32_t rs_swap_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
// sum over the 16 lanes of a DPP row (= the 16 tokens of a fragment), result in every lane, fixed order
__device__ __forceinline__ float rs_row16_sum(float v) {
    v += rs_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    v += rs_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    v += rs_dpp<0x141>(v);     // row_half_mirror
    v += rs_dpp<0x140>(v);     // row_mirror
    return v;
}

// LayerNorm backward on a D-layout fragment (see layer_norm_bwd): the gamma / beta gradient contributions of the 16 tokens
// are summed across the row at once and added to slot `vg` / `vb` (= lane n) of the compact accumulator.
template <int KT_>
__device__ __forceinline__ void rs_layer_norm_bwd(float (&g)[KT_][4], const float (&zh)[KT_][4], float rstd, const float* gam,
                                                  int g4, int n, int vg, int vb, float (&acc)[KT_][4]) {
    constexpr float invD = 1.0f / (16 * KT_);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int t = 0; t < KT_; ++t) {
        const float4 gm4 = *reinterpret_cast<const float4*>(gam + 16 * t + g4);
        const float gm[4] = {gm4.x, gm4.y, gm4.z, gm4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sg_ = rs_row16_sum(g[t][r] * zh[t][r]);
            const float sb_ = rs_row16_sum(g[t][r]);
            acc[t][r] += n == vg ? sg_ : (n == vb ? sb_ : 0.f);
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

// acc[MOFF+mt][NOFF+nt] += sum over the wave's 16 token rows of A[tok][16mt + .] * G[tok][16nt + .]   (mt < MT_, nt < NT_)
// al / gl: per-lane bases  buffer + (tile_row0 + 4 g) * ld + n ; step ks adds ONE row.  Which four rows form a contraction step
// is free (both operands use the same rows): lane group g takes row 4 g + ks, so the two groups of a 32-lane half read rows FOUR
// apart - with the row stride = 4 mod 8 that is 16 banks apart, conflict-free (rows one apart collide on 12 of 16 banks).
template <int MT_, int NT_, int MOFF, int NOFF, int LDA, int LDG, int MFULL, int NFULL>
__device__ __forceinline__ void rs_wgrad(const float* al, const float* gl, f32x4 (&acc)[MFULL][NFULL]) {
    float av[4][MT_], gv[4][NT_];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int mt = 0; mt < MT_; ++mt) av[ks][mt] = al[ks * LDA + 16 * mt];
#pragma unroll
        for (int nt = 0; nt < NT_; ++nt) gv[ks][nt] = gl[ks * LDG + 16 * nt];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT_; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT_; ++nt)
                acc[MOFF + mt][NOFF + nt] = mfma4(av[ks][mt], gv[ks][nt], acc[MOFF + mt][NOFF + nt]);
}

// chunk swizzle of a weight image: rows 4..11 of every 16 swap the two 16-byte chunks of each pair
__device__ __forceinline__ int rs_flip(int row) { return ((row & 15) >= 4 && (row & 15) < 12) ? 4 : 0; }

// global [R][C] -> swizzled LDS image with row stride ld (or its transpose), 16 bytes per load
__device__ __forceinline__ void rs_stage_image(const float* __restrict__ g, float* __restrict__ s, int R, int C, int ld,
                                               bool transpose) {
    const int c4n = C >> 2;
    for (int i = threadIdx.x; i < R * c4n; i += blockDim.x) {
        const int r = i / c4n, c = (i - r * c4n) << 2;
        const float4 v = *reinterpret_cast<const float4*>(g + (size_t)r * C + c);
        if (transpose) {   // image row = source column
            s[c * ld + (r ^ rs_flip(c))] = v.x; s[(c + 1) * ld + (r ^ rs_flip(c + 1))] = v.y;
            s[(c + 2) * ld + (r ^ rs_flip(c + 2))] = v.z; s[(c + 3) * ld + (r ^ rs_flip(c + 3))] = v.w;
        } else {
            *reinterpret_cast<float4*>(s + r * ld + (c ^ rs_flip(r))) = v;
        }
    }
}

// SAME: the Q and K roles share one generated-weight table (no 'pos' in the flag).  FT: the number of fields when it is known at
// compile time (row offsets, clamps and loop bounds of the attention phases then fold into immediates), 0 = read it from the
// descriptor.
template <int D, int U, int H, bool SAME, int FT>
__global__ __launch_bounds__(kRsBlock) void layer_bwd_rs_kernel(satrans_layer_desc a, int Tsamp, const float* __restrict__ dy,
                                                                 float* __restrict__ dx, float* __restrict__ slabs) {
    constexpr int KT = D / 16, UT = U / 16, d = D / H, LD = D + 4, LU = U + 4;
    constexpr int HB = (UT + KT - 1) / KT;          // row buffers needed to hold one U-wide operand (<= 2)
    constexpr int NB = (UT < KT) ? UT : KT;          // 16-feature tiles of such an operand held by one row buffer
    static_assert(HB <= 2 && UT == HB * NB, "MetaNet hidden width must be D/.. or 2*D for the fused backward");
    static_assert(KT <= 2, "the cached dropout keep flags hold 8 bits per site");
    static_assert(H <= 4, "one attention task per lane pair: T * H * F <= 64 * H <= 256");
    extern __shared__ __align__(16) float lds[];
    const int F = FT ? FT : a.F;
    const int Fh = (F + 1) >> 1;                     // keys [0, Fh) to lane 0 of a pair, [Fh, F) to lane 1
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Which two waves share a SIMD is the dispatcher's business: every wave reads its SIMD id (HW_ID bits 5:4) and the pairs
    // are formed from what the hardware did - the earlier wave of a SIMD takes the Q role, the later one the K role, and the
    // SIMD number is the pair's 16-token tile.  (256 registers per wave: exactly two of the eight waves on each of the four
    // SIMDs; should a SIMD ever report another count, the workgroup falls back to pairing wave w with wave w + 4.)
    int role = wave8 >> 2, tw = wave8 & 3;           // 0 = Q role (+ output block), 1 = K role (+ values); 16-token tile
    {
        const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;
        int* ids = (int*)lds;
        if (lane == 0) ids[wave8] = simd;
        __syncthreads();
        int rank = 0, per[4] = {0, 0, 0, 0};
        for (int w = 0; w < kRsWaves; ++w) {
            const int sw = ids[w];
            if (sw == simd && w < wave8) ++rank;
            for (int q = 0; q < 4; ++q) per[q] += sw == q;
        }
        const bool even = per[0] == 2 && per[1] == 2 && per[2] == 2 && per[3] == 2;
        if (even) {
            role = __builtin_amdgcn_readfirstlane(rank);
            tw = __builtin_amdgcn_readfirstlane(simd);
        }
        __syncthreads();                             // (the ids sit where the weight images are staged next)
    }
    const int n = lane & 15, g = lane >> 4, g4 = 4 * g;
    const bool meta_q = a.flags & SATRANS_META_Q, meta_k = a.flags & SATRANS_META_K;
    const bool meta_r = role ? meta_k : meta_q;      // is THIS wave's role modulated
    const bool relu_out = a.flags & SATRANS_RELU_OUT, use_res = !(a.flags & SATRANS_NO_RES);

    // ---- LDS: swizzled forward images, LN vectors, 5 row buffers, softmax statistics and caches (= K-role scratch) ------
    float* p = lds;
    auto take = [&](int cnt) { float* r = p; p += (cnt + 3) & ~3; return r; };
    float* wq = take(D * LD); float* wk = take(D * LD); float* wv = take(D * LD); float* woT = take(D * LD);
    float* w1q = take(D * LU); float* w2q = take(U * LD);
    float* w1k = SAME ? w1q : take(D * LU);
    float* w2k = SAME ? w2q : take(U * LD);
    float* lnq_g = take(D); float* lnk_g = take(D); float* ln_g = take(D);
    float* lnq_b = take(D); float* lnk_b = take(D); float* ln_b = take(D);
    float* sq = take(kRsRows * LD);     // q      | F: Q-role scratch
    float* sk = take(kRsRows * LD);     // k      -> dk
    float* sv = take(kRsRows * LD);     // v      -> dv
    float* so = take(kRsRows * LD);     // o      -> go | F: Q-role scratch
    float* sg = take(kRsRows * LD);     // du (phase C scratch) -> dq | F: Q-role scratch
    const int ntask_max = Tsamp * H * F;
    float* st_inv = take(ntask_max);                  // 1 / sum_j exp(s_ij - max_i)
    uint32_t* st_keep = (uint32_t*)take(ntask_max);   // bit j: attention-dropout keep flag of (i, j)   (F <= 32)
    const int cache1 = (ntask_max * F + 3) & ~3;
    const int cache_floats = max(2 * cache1, 3 * kRsRows * LD);
    float* sP = take(cache_floats);                   // exp(s_ij - max_i), the un-normalised softmax numerators
    float* sDS = sP + cache1;                         // dP_ij, then dS_ij (phases D, E)
    float* ka = sP;                                   // phase F: the K-role waves' three operand row buffers (caches are dead)
    float* kb = ka + kRsRows * LD;
    float* kc = kb + kRsRows * LD;
    const int stage_floats = (int)(p - sq);           // sq .. end of the caches: staging space of the final reductions

    const WorkRange wr = work_range(a.seg, a.S, Tsamp, gridDim.x, blockIdx.x);
    const bool idle = wr.g0 >= wr.g1;      // no tile for this workgroup: only its zero slab is due
    if (!idle) {
        rs_stage_image(a.w_query, wq, D, D, LD, false);
        rs_stage_image(a.w_key, wk, D, D, LD, false);
        rs_stage_image(a.w_value, wv, D, D, LD, false);
        rs_stage_image(a.w_out, woT, D, D, LD, true);      // woT[i][o] = Wo[o][i]  (nn.Linear: y = x @ Wo^T)
        for (int i = threadIdx.x; i < D; i += blockDim.x) {
            ln_g[i] = a.ln_g[i]; ln_b[i] = a.ln_b[i];
            if (meta_q) { lnq_g[i] = a.lnq_g[i]; lnq_b[i] = a.lnq_b[i]; }
            if (meta_k) { lnk_g[i] = a.lnk_g[i]; lnk_b[i] = a.lnk_b[i]; }
        }
        // rows of padding tokens are multiplied by exact zeros in the token-contraction products: they must hold finite
        // numbers from the start (0 * NaN would poison an accumulator)
        for (int i = threadIdx.x; i < stage_floats; i += blockDim.x) sq[i] = 0.f;
    }
    __syncthreads();

    // per-lane offsets into a swizzled image: by columns (chain: row 4g + r, column n) and by rows (chain_t: row n, column 4g)
    const int fl_g = (g == 1 || g == 2) ? 4 : 0, fl_n = (n >= 4 && n < 12) ? 1 : 0;
    const int lo_d = g4 * LD + (n ^ fl_g), lo_u = g4 * LU + (n ^ fl_g);
    const int lt_d = n * LD + 4 * (g ^ fl_n), lt_u = n * LU + 4 * (g ^ fl_n);
    // this wave's role: which images, LayerNorm vectors and dropout bits it works with
    const float* wp = role ? wk : wq;                  // projection of the role
    const float* w1 = role ? w1k : w1q;
    const float* w2 = role ? w2k : w2q;
    const float* lnm_g = role ? lnk_g : lnq_g;
    const float* lnm_b = role ? lnk_b : lnq_b;
    const int kshift = role ? 8 : 0;
    // the attention task of this lane pair - the same in every tile and every attention phase
    const int pair = (int)threadIdx.x >> 1, half = (int)threadIdx.x & 1;
    const int t0_ls = pair / (H * F), t0_rem = pair - t0_ls * H * F;
    const int t0_h = t0_rem / F, t0_i = t0_rem - t0_h * F;
    const int jb = half ? Fh : 0, je = half ? F : Fh;       // this lane's keys [jb, je)
    const FusedDrop dc = fused_drop(a);
    const float inv_sqrt_d = 1.0f / sqrtf((float)d);

    // ---- register accumulators of THIS role's weight gradients (whole kernel) ---------------------------------------------
    f32x4 acc_pa[KT][KT];      // Q role: dWq   K role: dWk
    f32x4 acc_pb[KT][KT];      // Q role: dWo   K role: dWv
    f32x4 acc_w1[KT][UT], acc_w2[UT][KT];
    float aln[KT][4];          // compact LayerNorm gradients: lane n = 0 / 1: MetaNet LN gamma / beta; 2 / 3 (Q role): output LN
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KT; ++i) {
#pragma unroll
        for (int j = 0; j < KT; ++j) { acc_pa[i][j] = zero4; acc_pb[i][j] = zero4; }
#pragma unroll
        for (int j = 0; j < UT; ++j) { acc_w1[i][j] = zero4; acc_w2[j][i] = zero4; }
#pragma unroll
        for (int r = 0; r < 4; ++r) aln[i][r] = 0.f;
    }

    // this wave's 16 rows of the row buffers
    const int row0 = 16 * tw;
    const int my = (row0 + n) * LD + g4;               // D-layout row access (float4 per t)
    const int wg = (row0 + g4) * LD + n;               // token-contraction access: lane group g reads rows 4g + ks
    // operand scratch of the role in phase F (own rows only): Q role the dead q / o / dq rows, K role the dead softmax caches
    float* s0 = role ? ka : sq;
    float* s1 = role ? kb : so;
    float* s2 = role ? kc : sg;

    // output: [G][CSZ] scenario-independent part, then [G + S][TSZ] generated-weight records (record index =
    // workgroup + scenario: strictly increasing along the global tile list, hence unique)
    constexpr int CSZ = 4 * D * D + 6 * D, TSZ = 4 * D * U;
    float* common = slabs + (size_t)blockIdx.x * CSZ;
    float* records = slabs + (size_t)gridDim.x * CSZ;
    float* stage = sq;
    using KTc = std::integral_constant<int, KT>;
    using UTc = std::integral_constant<int, UT>;
    // Combine the accumulators of one matrix over the waves that hold a share of it through LDS in a fixed order, write it out,
    // clear them.  who: 0 = nobody (zeros are written), 1 = the Q-role waves, 2 = the K-role waves, 3 = all eight.
    auto flush = [&](auto& acc, auto mtc, auto ntc, float* dst, int who) {
        constexpr int MT_ = decltype(mtc)::value, NT_ = decltype(ntc)::value;
        constexpr int ncols = 16 * NT_, sz = MT_ * 16 * ncols;
        const bool mine = who == 3 || who == role + 1;
        const int slot = who == 3 ? wave8 : tw, slots = who == 3 ? 8 : 4;
        if (mine) {
#pragma unroll
            for (int mt = 0; mt < MT_; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT_; ++nt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stage[slot * sz + (16 * mt + g4 + r) * ncols + 16 * nt + n] = acc[mt][nt][r];
                    acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < sz; e += kRsBlock) {
            float s = 0.f;
            if (who) {
                s = stage[e];
                for (int k = 1; k < slots; ++k) s += stage[k * sz + e];
            }
            dst[e] = s;
        }
        __syncthreads();
    };

    STAMP_DECL
    RSTAMP(8);
    int pre = 0;
    for (int scen = 0; scen < a.S && pre < wr.g1; ++scen) {
      const int nt_s = tiles_of(a.seg, scen, Tsamp);
      const int t0 = max(wr.g0, pre) - pre, t1 = min(wr.g1, pre + nt_s) - pre;
      pre += nt_s;
      if (t0 >= t1) continue;
      // ---- this scenario's generated MetaNet weights (the previous tile loop ended on a barrier) ----------------------------
      if (meta_q) {
          const float* row = a.tab_q + (size_t)scen * a.tab_stride;
          rs_stage_image(row, w1q, D, U, LU, false);
          rs_stage_image(row + D * U, w2q, U, D, LD, false);
      }
      if (meta_k && (!SAME || !meta_q)) {
          const float* row = a.tab_k + (size_t)scen * a.tab_stride;
          rs_stage_image(row, w1k, D, U, LU, false);
          rs_stage_image(row + D * U, w2k, U, D, LD, false);
      }
      __syncthreads();
      const int lo = a.seg[scen], hi = a.seg[scen + 1];
      // The sample index and the input row of a tile are fetched one tile ahead (at the end of phase F, when the registers
      // of the current x are free): nothing else would hide those two dependent global loads at the top of a tile.
      const int tok = row0 + n;
      const int ls_tok = tok / F, f_tok = tok - ls_tok * F;      // this lane's (sample, field) inside any tile that holds it
      int b_next = 0;
      float x_next[KT][4];
      auto fetch_tile = [&](int tile_) {
          const int first_ = lo + tile_ * Tsamp;
          const int ntok_ = min(Tsamp, hi - first_) * F;
          const bool has_ = tw < ((ntok_ + 15) >> 4);
          const bool valid_ = has_ && tok < ntok_;
          const int ls_ = valid_ ? ls_tok : 0, f_ = valid_ ? f_tok : 0;
          b_next = a.order[first_ + ls_];
          if (has_) load_frag<KT>(layer_x_row(a, b_next, f_, F, D) + g4, x_next);
      };
      fetch_tile(t0);
      for (int tile = t0; tile < t1; ++tile) {
        const int first = lo + tile * Tsamp;
        const int32_t* samp = a.order + first;
        const int nS = min(Tsamp, hi - first), ntok = nS * F, ntt = (ntok + 15) >> 4;
        const bool has_tile = tw < ntt;
        const bool valid = has_tile && tok < ntok;
        const int f = valid ? f_tok : 0;
        const int b = b_next;
        const size_t grow = ((size_t)b * F + f) * D + g4;       // this lane's row of dy / dx
        const bool task_ok = pair < nS * H * F;
        const int task = task_ok ? pair : 0;
        const int tls = task_ok ? t0_ls : 0, th = task_ok ? t0_h : 0, ti = task_ok ? t0_i : 0;

        // token-wise state of THIS role that lives from phase A to phase F
        float x[KT][4], in0[KT][4], hh[UT][4], zh[KT][4], dr[KT][4];
        float rstd_m = 0.f;
        // keep flags of this token lane at the role's MetaNet site (bits 0-7) and at the output site (bits 16-23), generated once
        uint32_t keepbits = 0xFFFFFFFFu;
        if (dc.on && has_tile) {
            keepbits = token_keep_bits<KT>(drop_sample_key(dc.key[role ? kSiteMetaK : kSiteMetaQ], (uint32_t)b), f, D, g4,
                                           dc.thresh) << kshift;
            if (role == 0)
                keepbits |= token_keep_bits<KT>(drop_sample_key(dc.key[kSiteOut], (uint32_t)b), f, D, g4, dc.thresh) << 16;
        }

        RSTAMP(0);
        // ================= phase A: forward chain of the role ======================================================
        if (has_tile) {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[t][r] = x_next[t][r];
            chain<KT, KT, LD>(wp + lo_d, x, in0);
            if (role) {
                float v[KT][4];
                chain<KT, KT, LD>(wv + lo_d, x, v);
                store_frag<KT>(sv + my, v);
            }
            float out[KT][4];
            if (meta_r) {
                float m[KT][4];
                chain<KT, UT, LU>(w1 + lo_u, in0, hh);
#pragma unroll
                for (int t = 0; t < UT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hh[t][r] = fmaxf(hh[t][r], 0.f);
                chain<UT, KT, LD>(w2 + lo_d, hh, m);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float mm = m[t][r];
                        if (dc.on) mm = (keepbits >> (kshift + 4 * t + r)) & 1u ? mm * dc.scale : 0.f;
                        m[t][r] = mm + in0[t][r];
                    }
                layer_norm_keep<KT>(m, zh, rstd_m);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const float4 gg = *reinterpret_cast<const float4*>(lnm_g + 16 * t + g4);
                    const float4 bb = *reinterpret_cast<const float4*>(lnm_b + 16 * t + g4);
                    out[t][0] = zh[t][0] * gg.x + bb.x; out[t][1] = zh[t][1] * gg.y + bb.y;
                    out[t][2] = zh[t][2] * gg.z + bb.z; out[t][3] = zh[t][3] * gg.w + bb.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[t][r] = in0[t][r];
            }
            store_frag<KT>((role ? sk : sq) + my, out);
        }
        rs_barrier();

        RSTAMP(1);
        // ================= phase B: attention forward; cache numerators, 1/sum and dropout keep bits ===================
        // A task = a pair of adjacent lanes; each takes half of the keys.  Scores are staged in the task's row of the numerator
        // cache (pre-scaled by log2(e)/sqrt(d)); slots past the lane's last key repeat its last key and are masked.
        if (task_ok) {
            const int tb = samp[tls];
            f32x2 qi[d / 2];
            load_row<d>(sq + (size_t)(tls * F + ti) * LD + th * d, qi);
            const float* kbase = sk + (size_t)(tls * F + jb) * LD + th * d;
            const float* vbase = sv + (size_t)(tls * F + jb) * LD + th * d;
            float* prow = sP + (size_t)task * F + jb;
            const int nk = je - jb;
            const float sc_scale = inv_sqrt_d * kLog2e;
            float mx = -INFINITY;
            auto chunk1 = [&](const int u0) {
                f32x2 kr[4][d / 2];
                float sc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) load_row<d>(kbase + (size_t)min(u0 + u, nk - 1) * LD, kr[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    sc[u] = dot_row<d>(qi, kr[u]) * sc_scale;
                    mx = fmaxf(mx, sc[u]);          // a padding slot repeats the last real score: the maximum is unchanged
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u0 + u < nk) prow[u0 + u] = sc[u];
            };
            if constexpr (FT != 0) {
#pragma unroll
                for (int u0 = 0; u0 < (FT + 1) / 2; u0 += 4) { chunk1(u0); __builtin_amdgcn_sched_barrier(0); }
            } else {
#pragma unroll 1
                for (int u0 = 0; u0 < nk; u0 += 4) chunk1(u0);
            }
            mx = fmaxf(mx, rs_swap(mx));
            // keep flags of this lane's keys: bit u = key jb + u
            uint32_t kw = 0xFFFFFFFFu;
            if (dc.on) {
                const uint32_t skey = drop_sample_key(dc.key[kSiteAttn], (uint32_t)tb);
                const uint32_t block0 = drop_attn_elem(th, F, ti, 0) >> 2;
                uint32_t bits = 0;
                if constexpr (FT != 0) {
                    constexpr int NBLK = (((FT + 1) / 2 - 1) >> 2) + 2;      // 4-key blocks a half of the keys can touch
                    const int blk0 = jb >> 2;
#pragma unroll
                    for (int c = 0; c < NBLK; ++c)
                        if (4 * (blk0 + c) < FT)
                            bits |= drop_keep4(skey, block0 + (uint32_t)(blk0 + c), dc.thresh) << (4 * c);
                    kw = bits >> (jb & 3);
                } else {
                    for (int blk = jb >> 2; 4 * blk < je; ++blk)
                        bits |= drop_keep4(skey, block0 + (uint32_t)blk, dc.thresh) << (4 * (blk - (jb >> 2)));
                    kw = bits >> (jb & 3);
                }
            }
            f32x2 oacc[d / 2];
#pragma unroll
            for (int e = 0; e < d / 2; ++e) oacc[e] = f32x2{0.f, 0.f};
            float sum = 0.f;
            auto chunk2 = [&](const int u0) {
                f32x2 vr[4][d / 2];
                float ex[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int uu = min(u0 + u, nk - 1);
                    ex[u] = prow[uu];
                    load_row<d>(vbase + (size_t)uu * LD, vr[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ex[u] = u0 + u < nk ? __builtin_amdgcn_exp2f(ex[u] - mx) : 0.f;
                    sum += ex[u];
                    float pe = ex[u];
                    if (dc.on) pe = (kw >> (u0 + u)) & 1u ? ex[u] * dc.scale : 0.f;
                    axpy_row<d>(pe, vr[u], oacc);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u0 + u < nk) prow[u0 + u] = ex[u];
            };
#pragma unroll 1
            for (int u0 = 0; u0 < nk; u0 += 4) chunk2(u0);
            sum += rs_swap(sum);
#pragma unroll
            for (int e = 0; e < d / 2; ++e) {
                oacc[e].x += rs_swap(oacc[e].x);
                oacc[e].y += rs_swap(oacc[e].y);
            }
            // the task's keep word (bit j = key j): this lane's bits moved to their place, the partner's OR-ed in
            uint32_t keep = dc.on ? (kw << jb) & (half ? ~0u << Fh : ~(~0u << Fh)) : 0xFFFFFFFFu;
            keep |= rs_swap_u(keep);
            const float inv = 1.0f / sum;
            if (half == 0) {
                st_inv[task] = inv;
                st_keep[task] = keep;
            }
            // each lane of the pair stores its half of the output row
            float* orow = so + (size_t)(tls * F + ti) * LD + th * d + (d / 2) * half;
#pragma unroll
            for (int e = 0; e < d / 4; ++e) {
                const f32x2 a0 = half ? oacc[d / 4 + e] : oacc[e];
                orow[2 * e] = a0.x * inv;
                orow[2 * e + 1] = a0.y * inv;
            }
        }
        rs_barrier();

        RSTAMP(2);
        // ================= phase C (Q role): output block forward + backward ======================================
        if (has_tile && role == 0) {
            float o[KT][4], u[KT][4], zo[KT][4], gy[KT][4];
            load_frag<KT>(so + my, o);
            chain<KT, KT, LD>(woT + lo_d, o, u);
            float keep[KT][4];      // multiplicative factor of du: dropout mask times ReLU mask
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
            float rstd_o;
            layer_norm_keep<KT>(u, zo, rstd_o);
            load_frag<KT>(dy + grow, gy, valid);
            rs_layer_norm_bwd<KT>(gy, zo, rstd_o, ln_g, g4, n, 2, 3, aln);          // gy is now dr
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dr[t][r] = use_res ? gy[t][r] : 0.f;
                    gy[t][r] *= keep[t][r];                                     // du
                }
            store_frag<KT>(sg + my, gy);                                        // du rows (zero for padding tokens)
            rs_wgrad<KT, KT, 0, 0, LD, LD>(sg + wg, so + wg, acc_pb);          // dWo[o][i] += du^T o
            float go[KT][4];
            chain_t<KT, KT, LD>(woT + lt_d, gy, go);                            // go = du Wo
            store_frag<KT>(so + my, go);
        }
        rs_barrier();

        RSTAMP(3);
        // ================= phase D: softmax backward by rows: dS_ij (cached for phase E) and dq_i =========================
        // pass 1: dP_ij = (go_i . v_j) * mask_ij staged in the task's row of the dS cache, dot_i = sum_j P_ij dP_ij;
        // pass 2: dS_ij = P_ij (dP_ij - dot_i) / sqrt(d) replaces it, the numerator cache row becomes P_ij * mask_ij
        //         (the coefficient of dv_j), dq_i = sum_j dS_ij k_j.  Each lane of a pair over its half of the keys.
        if (task_ok) {
            f32x2 gi[d / 2];
            load_row<d>(so + (size_t)(tls * F + ti) * LD + th * d, gi);
            const float* kbase = sk + (size_t)(tls * F + jb) * LD + th * d;
            const float* vbase = sv + (size_t)(tls * F + jb) * LD + th * d;
            float* prow = sP + (size_t)task * F + jb;
            float* drow = sDS + (size_t)task * F + jb;
            const int nk = je - jb;
            const float inv = st_inv[task];
            const uint32_t keep = st_keep[task] >> jb;
            const float scale = dc.scale;
            float dot = 0.f;
            auto chunk3 = [&](const int u0) {
                f32x2 vr[4][d / 2];
                float pj[4], dp[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int uu = min(u0 + u, nk - 1);
                    pj[u] = prow[uu];
                    load_row<d>(vbase + (size_t)uu * LD, vr[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    dp[u] = dot_row<d>(gi, vr[u]);
                    dp[u] = ((keep >> (u0 + u)) & 1u) ? dp[u] * scale : 0.f;
                    pj[u] = u0 + u < nk ? pj[u] * inv : 0.f;
                    dot = fmaf(pj[u], dp[u], dot);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u0 + u < nk) drow[u0 + u] = dp[u];
            };
            if constexpr (FT != 0) {
#pragma unroll
                for (int u0 = 0; u0 < (FT + 1) / 2; u0 += 4) { chunk3(u0); __builtin_amdgcn_sched_barrier(0); }
            } else {
#pragma unroll 1
                for (int u0 = 0; u0 < nk; u0 += 4) chunk3(u0);
            }
            dot += rs_swap(dot);
            f32x2 dq[d / 2];
#pragma unroll
            for (int e = 0; e < d / 2; ++e) dq[e] = f32x2{0.f, 0.f};
            auto chunk4 = [&](const int u0) {
                f32x2 kr[4][d / 2];
                float pj[4], ds[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int uu = min(u0 + u, nk - 1);
                    pj[u] = prow[uu];
                    ds[u] = drow[uu];
                    load_row<d>(kbase + (size_t)uu * LD, kr[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    pj[u] = u0 + u < nk ? pj[u] * inv : 0.f;
                    ds[u] = pj[u] * (ds[u] - dot) * inv_sqrt_d;
                    pj[u] = ((keep >> (u0 + u)) & 1u) ? pj[u] * scale : 0.f;
                    axpy_row<d>(ds[u], kr[u], dq);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u0 + u < nk) { drow[u0 + u] = ds[u]; prow[u0 + u] = pj[u]; }
            };
            if constexpr (FT != 0) {
#pragma unroll
                for (int u0 = 0; u0 < (FT + 1) / 2; u0 += 4) { chunk4(u0); __builtin_amdgcn_sched_barrier(0); }
            } else {
#pragma unroll 1
                for (int u0 = 0; u0 < nk; u0 += 4) chunk4(u0);
            }
            float* qrow = sg + (size_t)(tls * F + ti) * LD + th * d + (d / 2) * half;
#pragma unroll
            for (int e = 0; e < d / 4; ++e) {
                const float sx = dq[e].x + rs_swap(dq[e].x), sy = dq[e].y + rs_swap(dq[e].y);
                const float tx = dq[d / 4 + e].x + rs_swap(dq[d / 4 + e].x), ty = dq[d / 4 + e].y + rs_swap(dq[d / 4 + e].y);
                qrow[2 * e] = half ? tx : sx;
                qrow[2 * e + 1] = half ? ty : sy;
            }
        }
        rs_barrier();

        RSTAMP(4);
        // ================= phase E: by columns: lane 0 of a pair dk_j = sum_i dS_ij q_i, lane 1 dv_j = sum_i P_ij mask_ij go_i
        if (task_ok) {
            const int j = ti;
            f32x2 acc[d / 2];
#pragma unroll
            for (int e = 0; e < d / 2; ++e) acc[e] = f32x2{0.f, 0.f};
            const float* rbase = (half ? so : sq) + (size_t)(tls * F) * LD + th * d;      // go rows | q rows
            const float* col = (half ? sP : sDS) + (size_t)((tls * H + th) * F) * F + j;  // P mask | dS, column j
            auto chunk5 = [&](const int i0) {
                f32x2 rr[4][d / 2];
                float cf[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = min(i0 + u, F - 1);
                    cf[u] = col[(size_t)i * F];
                    load_row<d>(rbase + (size_t)i * LD, rr[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) axpy_row<d>(i0 + u < F ? cf[u] : 0.f, rr[u], acc);
            };
            if constexpr (FT != 0) {
#pragma unroll
                for (int i0 = 0; i0 < FT; i0 += 4) { chunk5(i0); __builtin_amdgcn_sched_barrier(0); }
            } else {
#pragma unroll 1
                for (int i0 = 0; i0 < F; i0 += 4) chunk5(i0);
            }
            store_row<d>((half ? sv : sk) + (size_t)(tls * F + j) * LD + th * d, acc, 1.0f);
        }
        rs_barrier();

        RSTAMP(5);
        // ================= phase F: MetaNet and projection backward of the role, its weight gradients, its part of dx =====
        if (has_tile) {
            float gout[KT][4];
            load_frag<KT>((role ? sk : sg) + my, gout, valid);      // gradient of the (post-MetaNet) keys / queries
            // rows >= ntok of sv still hold forward values: they are neutralised by x = 0 in the dWv product and masked when read
            // as a fragment

            if (meta_r) {                                            // submodules.py:77-103 backwards
                rs_layer_norm_bwd<KT>(gout, zh, rstd_m, lnm_g, g4, n, 0, 1, aln);            // gout = dz
                float dm[KT][4];
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float mk = 1.0f;
                        if (dc.on) mk = (keepbits >> (kshift + 4 * t + r)) & 1u ? dc.scale : 0.f;
                        dm[t][r] = gout[t][r] * mk;
                    }
                // dW2[u][o] += h^T dm : h goes to scratch 0 (and 1), dm to scratch 2
                {
                    float part[KT][4];
#pragma unroll
                    for (int t = 0; t < KT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) part[t][r] = (t < NB) ? hh[t % UT][r] : 0.f;
                    store_frag<KT>(s0 + my, part);
                    if constexpr (HB == 2) {
#pragma unroll
                        for (int t = 0; t < KT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) part[t][r] = (t < NB) ? hh[(NB + t) % UT][r] : 0.f;
                        store_frag<KT>(s1 + my, part);
                    }
                    store_frag<KT>(s2 + my, dm);
                    rs_wgrad<NB, KT, 0, 0, LD, LD>(s0 + wg, s2 + wg, acc_w2);
                    if constexpr (HB == 2) rs_wgrad<NB, KT, NB, 0, LD, LD>(s1 + wg, s2 + wg, acc_w2);
                }
                // dh = (dm W2^T) * [h > 0]
                float dh[UT][4];
                chain_t<KT, UT, LD>(w2 + lt_d, dm, dh);
#pragma unroll
                for (int t = 0; t < UT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dh[t][r] = hh[t][r] > 0.f ? dh[t][r] : 0.f;
                // dW1[i][u] += in0^T dh : in0 to scratch 2, dh to scratch 0 (and 1)
                {
                    float part[KT][4];
#pragma unroll
                    for (int t = 0; t < KT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) part[t][r] = (t < NB) ? dh[t % UT][r] : 0.f;
                    store_frag<KT>(s0 + my, part);
                    if constexpr (HB == 2) {
#pragma unroll
                        for (int t = 0; t < KT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) part[t][r] = (t < NB) ? dh[(NB + t) % UT][r] : 0.f;
                        store_frag<KT>(s1 + my, part);
                    }
                    store_frag<KT>(s2 + my, in0, valid);
                    rs_wgrad<KT, NB, 0, 0, LD, LD>(s2 + wg, s0 + wg, acc_w1);
                    if constexpr (HB == 2) rs_wgrad<KT, NB, 0, NB, LD, LD>(s2 + wg, s1 + wg, acc_w1);
                }
                // gradient of the MetaNet input: dz + dh W1^T
                float back[KT][4];
                chain_t<UT, KT, LU>(w1 + lt_u, dh, back);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gout[t][r] += back[t][r];
            }

            // projection of the role: dW[i][o] += x^T g ; dx part = g W^T
            store_frag<KT>(s0 + my, x, valid);
            store_frag<KT>(s1 + my, gout);
            rs_wgrad<KT, KT, 0, 0, LD, LD>(s0 + wg, s1 + wg, acc_pa);
            float back[KT][4];
            chain_t<KT, KT, LD>(wp + lt_d, gout, back);
            if (role == 0) {
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dr[t][r] += back[t][r];
            } else {
                rs_wgrad<KT, KT, 0, 0, LD, LD>(s0 + wg, sv + wg, acc_pb);         // dWv += x^T dv
                float gv[KT][4], bv[KT][4];
                load_frag<KT>(sv + my, gv, valid);
                chain_t<KT, KT, LD>(wv + lt_d, gv, bv);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) back[t][r] += bv[t][r];
                store_frag<KT>(s2 + my, back);                                      // the K role's part of dx, for its Q partner
            }
        }
        if (tile + 1 < t1) fetch_tile(tile + 1);
        rs_barrier();
        RSTAMP(6);
        // dx = dr + gq Wq^T + (gk Wk^T + gv Wv^T): the Q-role wave adds its partner's part (the K scratch is next written in
        // phase B of the following tile, two barriers from here)
        if (has_tile && role == 0) {
            float part[KT][4];
            load_frag<KT>(kc + my, part);
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dr[t][r] += part[t][r];
            if (valid) store_frag<KT>(dx + grow, dr);
        }
      }
      // ---- this scenario's generated-weight gradients: record (workgroup + scenario) ---------------------------------
      // with one shared table both roles' waves hold shares of the same two matrices; the reducer reads the part of a role only
      // when that role is active, so the sums go to the Q part when Q is modulated, else to the K part
      {
          __syncthreads();                              // (the last tile's dx read of the K scratch is inside the staging area)
          float* rec = records + (size_t)(blockIdx.x + scen) * TSZ;
          if constexpr (SAME) {
              const bool to_k = !meta_q;
              flush(acc_w1, KTc{}, UTc{}, rec + (to_k ? 2 * D * U : 0), 3);
              flush(acc_w2, UTc{}, KTc{}, rec + (to_k ? 3 * D * U : D * U), 3);
              flush(acc_w1, KTc{}, UTc{}, rec + (to_k ? 0 : 2 * D * U), 0);
              flush(acc_w2, UTc{}, KTc{}, rec + (to_k ? D * U : 3 * D * U), 0);
          } else {
              flush(acc_w1, KTc{}, UTc{}, rec, 1);
              flush(acc_w2, UTc{}, KTc{}, rec + D * U, 1);
              flush(acc_w1, KTc{}, UTc{}, rec + 2 * D * U, 2);
              flush(acc_w2, UTc{}, KTc{}, rec + 3 * D * U, 2);
          }
      }
    }

    RSTAMP(7);
    // ---- scenario-independent gradients of this workgroup ------------------------------------------------------------------
    __syncthreads();
    flush(acc_pa, KTc{}, KTc{}, common, 1);                   // dWq: Q-role waves
    flush(acc_pa, KTc{}, KTc{}, common + D * D, 2);           // dWk: K-role waves
    flush(acc_pb, KTc{}, KTc{}, common + 2 * D * D, 2);       // dWv
    flush(acc_pb, KTc{}, KTc{}, common + 3 * D * D, 1);       // dWo
    // LayerNorm vectors: lane n = vector id; [wave8][4 vectors][D] through LDS, then the four waves of the role in order
    if (n < 4) {
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(wave8 * 4 + n) * D + 16 * t + g4 + r] = aln[t][r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 6 * D; e += kRsBlock) {
        const int v = e / D, f = e - v * D;          // [ln g | ln b | lnq g | lnq b | lnk g | lnk b]
        const int rl = v >= 4 ? 1 : 0, vid = v < 2 ? 2 + v : (v < 4 ? v - 2 : v - 4);
        float s = stage[((rl * 4) * 4 + vid) * D + f];
        for (int w = 1; w < 4; ++w) s += stage[((rl * 4 + w) * 4 + vid) * D + f];
        common[4 * D * D + e] = s;
    }
}

// -------------------------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------------------------
static int64_t rs_lds_floats(int T, int F, int D, int U, int H, bool same_tab) {
    const int LD = D + 4, LU = U + 4;
    auto r4 = [](int64_t v) { return (v + 3) & ~(int64_t)3; };
    const int64_t tasks = (int64_t)T * H * F;
    const int64_t cache = std::max<int64_t>(2 * r4(tasks * F), 3 * (int64_t)kRsRows * LD);
    return 4 * (int64_t)D * LD + (same_tab ? 1 : 2) * ((int64_t)D * LU + (int64_t)U * LD) + 6 * D + 5 * kRsRows * LD +
           2 * r4(tasks) + cache + 64;
}

struct RsPlan {
    int T, G;
    size_t lds;
};

static bool rs_plan(const satrans_layer_desc* d, RsPlan& p) {
    if (!d || d->F < 1 || d->F > 32) return false;                  // keep bits: one 32-bit word per row
    if (d->flags & (SATRANS_GATE | SATRANS_BILINEAR)) return false;
    const bool meta = d->flags & (SATRANS_META_Q | SATRANS_META_K);
    const bool shape = (d->D == 32 && d->H == 4 && (!meta || d->U == 64)) || (d->D == 16 && d->H == 2 && (!meta || d->U == 32));
    if (!shape) return false;
    const bool same_tab = d->tab_q == d->tab_k;
    p.T = 64 / d->F;
    p.lds = (size_t)rs_lds_floats(p.T, d->F, d->D, d->U, d->H, same_tab) * 4;
    if (p.lds > 160 * 1024) return false;
    // staging space of the final reductions: eight shares of the largest matrix (D x U)
    const int LD = d->D + 4;
    const int64_t tasks = (int64_t)p.T * d->H * d->F;
    const int64_t stage = 5 * kRsRows * LD + 2 * ((tasks + 3) & ~3) +
                          std::max<int64_t>(2 * ((tasks * d->F + 3) & ~3), 3 * (int64_t)kRsRows * LD);
    if (stage < 8 * (int64_t)d->D * std::max(d->U, d->D)) return false;
    const int64_t tiles = ceil_div(d->B, p.T) + d->S;
    p.G = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, cu_count()));      // one workgroup per CU, one round
    return true;
}

template <int D, int U, int H, bool SAME, int FT>
static int rs_launch(const satrans_layer_desc* d, const RsPlan& p, const float* dy, float* dx, float* slabs, hipStream_t stream) {
    static size_t attr_set = 0;
    if (p.lds > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)layer_bwd_rs_kernel<D, U, H, SAME, FT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "layer_bwd(role-split): LDS attribute: %s", hipGetErrorString(e));
        attr_set = p.lds;
    }
    layer_bwd_rs_kernel<D, U, H, SAME, FT><<<p.G, kRsBlock, p.lds, stream>>>(*d, p.T, dy, dx, slabs);
    SATRANS_CHECK_LAUNCH("layer_bwd_rs_kernel");
    return SATRANS_OK;
}

}  // namespace satrans

using namespace satrans;

static int g_rs_enabled = -1;      // -1: read SATRANS_BWD_RS once (default on)

extern "C" int satrans_set_layer_bwd_rs(int on) {
    g_rs_enabled = on < 0 ? -1 : (on != 0);
    return SATRANS_OK;
}

extern "C" int satrans_layer_bwd_rs_supported(const satrans_layer_desc* d) {
    if (g_rs_enabled < 0) g_rs_enabled = getenv("SATRANS_BWD_RS") ? (atoi(getenv("SATRANS_BWD_RS")) != 0) : 1;
    RsPlan p;
    return g_rs_enabled && rs_plan(d, p) ? 1 : 0;
}

// same slab layout as layer_bwd_fused_kernel: [G][CSZ] + [G + S][TSZ]
extern "C" int64_t satrans_layer_bwd_rs_slab_floats(const satrans_layer_desc* d) {
    RsPlan p;
    if (!rs_plan(d, p)) return -1;
    const int64_t CSZ = 4 * (int64_t)d->D * d->D + 6 * d->D, TSZ = 4 * (int64_t)d->D * d->U;
    return (int64_t)p.G * CSZ + (int64_t)(p.G + d->S) * TSZ;
}

extern "C" int satrans_layer_bwd_rs_launch(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, int* T_out,
                                           int* G_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    RsPlan p;
    SATRANS_REQUIRE(rs_plan(d, p), SATRANS_E_UNSUPPORTED, "layer_bwd(role-split): shape not built");
    *T_out = p.T;
    *G_out = p.G;
    const bool same = d->tab_q == d->tab_k;
    static const bool f_const = !(getenv("SATRANS_BWD_FCONST") && atoi(getenv("SATRANS_BWD_FCONST")) == 0);
    if (d->D == 32) {
        if (same && d->F == 19 && f_const) return rs_launch<32, 64, 4, true, 19>(d, p, dy, dx, slabs, stream);
        return same ? rs_launch<32, 64, 4, true, 0>(d, p, dy, dx, slabs, stream)
                    : rs_launch<32, 64, 4, false, 0>(d, p, dy, dx, slabs, stream);
    }
    return same ? rs_launch<16, 32, 2, true, 0>(d, p, dy, dx, slabs, stream)
                : rs_launch<16, 32, 2, false, 0>(d, p, dy, dx, slabs, stream);
}

// Diagnostic: where the dispatcher puts the eight 256-register waves of a workgroup.  out[block * 8 + wave] = HW_ID
// (bits 3:0 wave slot, 5:4 SIMD, 11:8 CU, 15:13 SE) of that wave.
__global__ __launch_bounds__(satrans::kRsBlock) void rs_wave_map_kernel(int32_t* out) {
    extern __shared__ __align__(16) float lds[];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");              // 256 registers per wave, as the real kernel
    const int hw = (int)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
    if ((threadIdx.x & 63) == 0) {
        lds[threadIdx.x >> 6] = 0.f;
        out[blockIdx.x * satrans::kRsWaves + (threadIdx.x >> 6)] = hw;
    }
}

extern "C" int satrans_debug_wave_map(int32_t* out, int blocks, int lds_bytes, void* stream) {
    hipFuncSetAttribute((const void*)rs_wave_map_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    rs_wave_map_kernel<<<blocks, satrans::kRsBlock, lds_bytes, (hipStream_t)stream>>>(out);
    SATRANS_CHECK_LAUNCH("rs_wave_map_kernel");
    return SATRANS_OK;
}

#ifdef SATRANS_STAMPS
extern "C" int satrans_debug_read_stamps_rs(unsigned long long* h_out, int reset) {
    if (hipMemcpyFromSymbol(h_out, HIP_SYMBOL(satrans::g_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(satrans::g_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
