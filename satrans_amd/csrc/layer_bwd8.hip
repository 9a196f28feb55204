// Meta_Transformer_Layer backward, 8-wave version: two waves per SIMD.
//
// Reference: models/satrans.py:50-100 (layer), models/submodules.py:77-103 (MetaNet).  Same mathematics, inputs, outputs and
// slab layout as layer_bwd_fused_kernel (layer_fused.hip); what differs is how the work sits on the CU:
//
//   * one workgroup of EIGHT waves per CU = two HALVES of four waves that share one copy of the weight images in LDS but work
//     on DIFFERENT tiles (up to 64 tokens each: T samples, T*F <= 64) with their own row buffers and their own barrier (an LDS
//     counter: s_barrier would couple them).  Two waves share every SIMD and - because the halves drift apart - are usually
//     in different phases: the MFMA-bound token phases of one half run underneath the VALU / LDS-bound attention phases of
//     the other (the 4-wave kernel holds a full copy of every weight-gradient accumulator per wave - 128 registers - and is
//     stuck at one wave per SIMD with every latency exposed; a first version of this kernel ran all eight waves through the
//     phases in lockstep on 128-token tiles and was no faster: 290 us against 279 us per launch);
//   * the weight-gradient accumulators are SPLIT inside a half: of an (MT x NT)-tile gradient matrix every wave owns one or two
//     16x16 output tiles (or one tile over a share of the token rows when there are fewer than four tiles) and contracts them
//     over the token rows of ALL four waves: 32 accumulator registers per wave instead of 128.  The token-row operands of those
//     products (h, dm, in0, dh, x, gq, gk, dv, du, o) are exchanged through the half's six row buffers in three rounds per tile;
//   * LayerNorm gamma / beta gradients are reduced over the 16 token lanes with DPP adds as they are produced and kept in
//     8 registers (lane n of a 16-lane row holds vector n) instead of 48;
//   * nothing F x F is cached: the softmax backward recomputes P_ij from q_i, k_j and the row statistics (max, 1/sum) of
//     the forward pass, and uses dot_i = sum_j P_ij dP_ij = go_i . o_i (computed in the token phase), so the row pass
//     (dq_i) and the column pass (dk_i, dv_i) of a lane need no data from other lanes' passes and run back to back without
//     a barrier.  Pure VALU work instead of 35 KB of LDS per 64 tokens;
//   * fixed summation orders everywhere (no float atomics): bitwise reproducible from run to run.
//
// Phases per tile (the four waves of a half; || = barrier of the half):
//   A  token   forward chain x -> q0,k0,v -> MetaNet(q0), MetaNet(k0) -> q,k,v rows                                  ||
//   B  task    attention forward: o_i, row statistics (max, 1/sum), dropout keep word                               ||
//   C  token   Out_linear + residual + LayerNorm forward/backward -> dr, du rows, go rows, dot_i                    ||
//   D  split   dWo += du^T o ; task: dq_i (row pass), dk_i, dv_i (column pass) in registers                         ||
//      task    write dq, dk, dv rows                                                                              ||
//   F  token   MetaNet backward (Q, K), projections backward, dx; split weight gradients in three exchange rounds  (6 ||)
#include "layer_fused_common.h"

namespace satrans {

constexpr int kB8Waves = 8;
constexpr int kB8Block = 64 * kB8Waves;
constexpr int kHalfWaves = 4;
constexpr int kB8Rows = 16 * kHalfWaves;    // token rows of a half's tile


// Workgroup barrier for LDS hand-offs inside the tile loop.  __syncthreads() makes hipcc wait for EVERY outstanding memory
// operation (s_waitcnt vmcnt(0)) in front of s_barrier, which would stall all eight waves on the global prefetches that are
// meant to fly across the phases (next tile's sample index and input row, dy); only the LDS traffic has to be complete here.
// Global memory is never used to pass data between waves inside the loop.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Barrier of ONE half (four waves) of the workgroup: a monotonic arrival counter in LDS.  Every wave adds one after its own LDS
// traffic has completed, then polls until all four arrivals of this round are in (`round` counts this wave's barriers).  The
// counter never wraps in practice (4 per barrier, a few thousand barriers per launch).  All four waves of a half execute the
// same sequence of barriers by construction (the phases branch inside, never around, a barrier).
__device__ __forceinline__ void half_barrier(unsigned* ctr, unsigned& round) {
    round += kHalfWaves;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < round) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (= the 16 tokens of a fragment), result in every lane, fixed order
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    v += dpp_move<0x140>(v);     // row_mirror
    return v;
}

// LayerNorm backward on a D-layout fragment (see layer_norm_bwd): the gamma / beta gradient contributions of the 16 tokens
// are summed across the row at once and added to slot `vg` / `vb` (= lane n) of the compact accumulator.
template <int KT_>
__device__ __forceinline__ void layer_norm_bwd_c(float (&g)[KT_][4], const float (&zh)[KT_][4], float rstd, const float* gam,
                                                 int g4, int n, int vg, int vb, float (&acc)[KT_][4]) {
    constexpr float invD = 1.0f / (16 * KT_);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int t = 0; t < KT_; ++t) {
        const float4 gm4 = *reinterpret_cast<const float4*>(gam + 16 * t + g4);
        const float gm[4] = {gm4.x, gm4.y, gm4.z, gm4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sg_ = row16_sum(g[t][r] * zh[t][r]);
            const float sb_ = row16_sum(g[t][r]);
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

// One wave's share of a weight-gradient product  dW[16 mt + i][16 nt + j] += sum_rows A[row][16 mt + i] * G[row][16 nt + j]  over
// the 64 token rows of its half.  The MT x NT output tiles (TILES = 1, 2, 4 or 8) are dealt to the four waves: with 8 tiles a
// wave owns tiles w and w + 4 (SLOTS = 2), with 4 tiles one each; with fewer, KS = 4 / TILES waves share a tile and split the
// rows.  An operand wider than one row buffer (the U-wide hidden rows) spans two buffers: its first NB* tiles in X0, the rest in X1.
template <int MT, int NT>
struct SplitShape {
    static constexpr int TILES = MT * NT;
    static_assert(TILES == 1 || TILES == 2 || TILES == 4 || TILES == 8, "tiles per gradient matrix");
    static constexpr int SLOTS = TILES > kHalfWaves ? TILES / kHalfWaves : 1;
    static constexpr int KS = TILES >= kHalfWaves ? 1 : kHalfWaves / TILES;
    static constexpr int NK = (kB8Rows / 4) / KS;      // k-steps (4 token rows each) of one share
};
template <int MT, int NT, int NBA, int NBG, int LDX>
__device__ __forceinline__ void wgrad_split(const float* A0, const float* A1, const float* G0, const float* G1, int wave, int n,
                                            int g, f32x4 (&acc)[SplitShape<MT, NT>::SLOTS]) {
    using SS = SplitShape<MT, NT>;
    constexpr int TILES = SS::TILES, NK = SS::NK;
#pragma unroll
    for (int slot = 0; slot < SS::SLOTS; ++slot) {
        const int tile = TILES >= kHalfWaves ? wave + kHalfWaves * slot : wave % TILES;
        const int kpart = TILES >= kHalfWaves ? 0 : wave / TILES;
        const int mt = tile / NT, nt = tile - mt * NT;
        // Which four token rows form a contraction step is free (both operands use the same rows): lane group g of step ks takes
        // row 16 (ks / 4) + 4 g + ks % 4 of the share, so that the two groups of a 32-lane half read rows FOUR apart - with the
        // row stride LDX = 4 mod 8 that is 16 banks apart, conflict-free (rows one apart collide on 12 of 16 banks)
        const float* al = (mt < NBA ? A0 + 16 * mt : A1 + 16 * (mt - NBA)) + (kpart * NK * 4 + 4 * g) * LDX + n;
        const float* gl = (nt < NBG ? G0 + 16 * nt : G1 + 16 * (nt - NBG)) + (kpart * NK * 4 + 4 * g) * LDX + n;
        constexpr int CH = NK < 8 ? NK : 8;
#pragma unroll
        for (int k0 = 0; k0 < NK; k0 += CH) {
            float av[CH], gv[CH];
#pragma unroll
            for (int ks = 0; ks < CH; ++ks) {
                av[ks] = al[(16 * ((k0 + ks) >> 2) + ((k0 + ks) & 3)) * LDX];
                gv[ks] = gl[(16 * ((k0 + ks) >> 2) + ((k0 + ks) & 3)) * LDX];
            }
#pragma unroll
            for (int ks = 0; ks < CH; ++ks) acc[slot] = mfma4(av[ks], gv[ks], acc[slot]);
        }
    }
}

// SAME: the Q and K roles share one generated-weight table (no 'pos' in the flag).  FC: 4-key chunks of a score row (4 FC >= F).
// FT: the number of fields when it is known at compile time (every row offset, clamp and loop bound of the attention phases
// then folds into an immediate), 0 = read it from the descriptor.
template <int D, int U, int H, bool SAME, int FC, int FT>
__global__ __launch_bounds__(kB8Block) void layer_bwd8_kernel(satrans_layer_desc a, int Tsamp, const float* __restrict__ dy,
                                                               float* __restrict__ dx, float* __restrict__ slabs) {
    constexpr int KT = D / 16, UT = U / 16, d = D / H, LD = D + 4, LU = U + 4;
    static_assert(UT == 2 * KT, "the hidden rows span exactly two row buffers (U = 2 D)");
    static_assert(KT <= 2, "the cached dropout keep flags hold 8 bits per site");
    static_assert(d == 8, "head dimension 8 (two lane groups of a token share a head)");
    static_assert(H <= 4, "one attention task per thread of a half: T * H * F <= 64 * H <= 256");
    extern __shared__ __align__(16) float lds[];
    const int F = FT ? FT : a.F;
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave8 >> 2, wave = wave8 & 3;          // which half of the workgroup, wave inside the half
    const int tid_h = (int)threadIdx.x & 255;               // thread inside the half
    const int n = lane & 15, g = lane >> 4, g4 = 4 * g;
    const bool meta_q = a.flags & SATRANS_META_Q, meta_k = a.flags & SATRANS_META_K;
    const bool relu_out = a.flags & SATRANS_RELU_OUT, use_res = !(a.flags & SATRANS_NO_RES);

    // ---- LDS: forward weight images (read by rows for the transposed products), LN vectors, six row buffers, row statistics
    float* p = lds;
    auto take = [&](int cnt) { float* r = p; p += (cnt + 3) & ~3; return r; };
    float* wq = take(D * LD); float* wk = take(D * LD); float* wv = take(D * LD); float* woT = take(D * LD);
    float* w1q = take(D * LU); float* w2q = take(U * LD);
    float* w1k = SAME ? w1q : take(D * LU);
    float* w2k = SAME ? w2q : take(U * LD);
    float* lnq_g = take(D); float* lnk_g = take(D); float* ln_g = take(D);
    float* lnq_b = take(D); float* lnk_b = take(D); float* ln_b = take(D);
    float* bufs = take(2 * 6 * kB8Rows * LD);          // six row buffers per half, back to back
    float* sq = bufs + (size_t)half * 6 * kB8Rows * LD; // q            | F: exchange E0
    float* sk = sq + kB8Rows * LD;                      // k  -> dk     | F: exchange E4
    float* sv = sk + kB8Rows * LD;                      // v  -> dv     | F: exchange (after dWv)
    float* so = sv + kB8Rows * LD;                      // o            | F: exchange E1
    float* x1 = so + kB8Rows * LD;                      // du -> dq     | F: exchange E3
    float* x2 = x1 + kB8Rows * LD;                      // go           | F: exchange E2
    const int ntask_max = Tsamp * H * F;
    float4* st = (float4*)take(2 * 4 * ntask_max) + (size_t)half * ntask_max;   // per (sample, head, row): max, 1/sum, dot, keep word
    unsigned* hb_ctr = (unsigned*)take(4) + half;       // arrival counter of this half's barrier
    unsigned hb_round = 0;

    const WorkRange wr = work_range(a.seg, a.S, Tsamp, gridDim.x, blockIdx.x);
    const bool idle = wr.g0 >= wr.g1;      // no tile for this workgroup: only its zero slab is due
    if (!idle) {
        stage_image(a.w_query, wq, D, D, LD, false);
        stage_image(a.w_key, wk, D, D, LD, false);
        stage_image(a.w_value, wv, D, D, LD, false);
        stage_image(a.w_out, woT, D, D, LD, true);
        for (int i = threadIdx.x; i < D; i += blockDim.x) {
            ln_g[i] = a.ln_g[i]; ln_b[i] = a.ln_b[i];
            if (meta_q) { lnq_g[i] = a.lnq_g[i]; lnq_b[i] = a.lnq_b[i]; }
            if (meta_k) { lnk_g[i] = a.lnk_g[i]; lnk_b[i] = a.lnk_b[i]; }
        }
        // rows of padding tokens meet exact zeros in the token-contraction products: they must hold finite numbers from the
        // start (0 * NaN would poison an accumulator)
        for (int i = threadIdx.x; i < 2 * 6 * kB8Rows * LD; i += blockDim.x) bufs[i] = 0.f;
    }
    if (threadIdx.x < 2) (hb_ctr - half)[threadIdx.x] = 0u;
    __syncthreads();

    const int lo_d = g4 * LD + n, lo_u = g4 * LU + n;          // per-lane offset into an image: row 4g, column n
    const int lt_d = n * LD + g4, lt_u = n * LU + g4;          // ... for a read by rows (chain_t): row n, column 4g
    // (sample, head, row) of this thread's attention task - the same in every tile
    const int t0_ls = tid_h / (H * F), t0_rem = tid_h - t0_ls * H * F;
    const int t0_h = t0_rem / F, t0_i = t0_rem - t0_h * F;
    const FusedDrop dc = fused_drop(a);
    const float inv_sqrt_d = 1.0f / sqrtf((float)d);
    const float sc_scale = inv_sqrt_d * kLog2e;

    // ---- this wave's accumulators: one output tile (or tile share) of every gradient matrix; compact LN gradients ------
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int SP = SplitShape<KT, KT>::SLOTS, S1 = SplitShape<KT, UT>::SLOTS;
    f32x4 acc_wq[SP], acc_wk[SP], acc_wv[SP], acc_wo[SP];
    f32x4 acc_w1q[S1], acc_w2q[S1], acc_w1k[S1], acc_w2k[S1];
#pragma unroll
    for (int i = 0; i < SP; ++i) { acc_wq[i] = zero4; acc_wk[i] = zero4; acc_wv[i] = zero4; acc_wo[i] = zero4; }
#pragma unroll
    for (int i = 0; i < S1; ++i) { acc_w1q[i] = zero4; acc_w2q[i] = zero4; acc_w1k[i] = zero4; acc_w2k[i] = zero4; }
    float aln[KT][4];       // lane n: 0 = ln gamma, 1 = ln beta, 2 = lnq gamma, 3 = lnq beta, 4 = lnk gamma, 5 = lnk beta
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) aln[t][r] = 0.f;

    // this wave's 16 token rows of every row buffer: D-layout row access (float4 per t)
    const int row0 = 16 * wave;
    const int my = (row0 + n) * LD + g4;

    constexpr int CSZ = 4 * D * D + 6 * D, TSZ = 4 * D * U;
    float* common = slabs + (size_t)blockIdx.x * CSZ;
    float* records = slabs + (size_t)gridDim.x * CSZ;
    float* stage = bufs;   // the twelve row buffers are contiguous
    // combine the shares of one matrix - two halves x KS row shares per tile - through LDS in (half, share) order, write it out,
    // clear the accumulators
    auto flush = [&](auto& acc, auto mtc, auto ntc, float* dst, bool live) {
        constexpr int MT_ = decltype(mtc)::value, NT_ = decltype(ntc)::value;
        using SS = SplitShape<MT_, NT_>;
        constexpr int TILES = SS::TILES, ncols = 16 * NT_, PARTS = 2 * SS::KS;
        if (live) {
#pragma unroll
            for (int slot = 0; slot < SS::SLOTS; ++slot) {
                const int tile = TILES >= kHalfWaves ? wave + kHalfWaves * slot : wave % TILES;
                const int kpart = TILES >= kHalfWaves ? 0 : wave / TILES;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    stage[((half * SS::KS + kpart) * TILES + tile) * 256 + (g4 + r) * 16 + n] = acc[slot][r];
                acc[slot] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < TILES * 256; e += kB8Block) {
            const int tile = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
            const int mt = tile / NT_, nt = tile - mt * NT_;
            float s = 0.f;
            if (live) {
                s = stage[e];
#pragma unroll
                for (int k = 1; k < PARTS; ++k) s += stage[k * TILES * 256 + e];
            }
            dst[(16 * mt + rr) * ncols + 16 * nt + cc] = s;
        }
        __syncthreads();
    };
    using KTc = std::integral_constant<int, KT>;
    using UTc = std::integral_constant<int, UT>;

    STAMP_DECL
    STAMP(13);
    int pre = 0;
    for (int scen = 0; scen < a.S && pre < wr.g1; ++scen) {
      const int nt_s = tiles_of(a.seg, scen, Tsamp);
      const int t0 = max(wr.g0, pre) - pre, t1 = min(wr.g1, pre + nt_s) - pre;
      pre += nt_s;
      if (t0 >= t1) continue;
      // ---- this scenario's generated MetaNet weights (the previous tile loop ended on a barrier) ----------------------
      if (meta_q) {
          const float* row = a.tab_q + (size_t)scen * a.tab_stride;
          stage_image(row, w1q, D, U, LU, false);
          stage_image(row + D * U, w2q, U, D, LD, false);
      }
      if (meta_k && (!SAME || !meta_q)) {
          const float* row = a.tab_k + (size_t)scen * a.tab_stride;
          stage_image(row, w1k, D, U, LU, false);
          stage_image(row + D * U, w2k, U, D, LD, false);
      }
      __syncthreads();
      const int lo = a.seg[scen], hi = a.seg[scen + 1];
      const int tok = row0 + n;
      const int ls_tok = tok / F, f_tok = tok - ls_tok * F;      // this lane's (sample, field) inside any tile that holds it
      // The sample index and the input row of a tile are fetched one tile ahead (at the end of phase F): the eight waves run
      // the phases in lockstep, so nothing else would hide those two dependent global loads at the top of a tile.
      int b_next = 0;
      float x_next[KT][4];
      auto fetch_tile = [&](int tile_) {
          const int first_ = lo + tile_ * Tsamp;
          const bool valid_ = tok < min(Tsamp, hi - first_) * F;
          b_next = a.order[first_ + (valid_ ? ls_tok : 0)];
          load_frag<KT>(layer_x_row(a, b_next, valid_ ? f_tok : 0, F, D) + g4, x_next, valid_);
      };
#ifdef SATRANS_ONE_HALF      // diagnostic build: half 0 takes every tile, half 1 idles (how long does ONE wave per SIMD take?)
      const int t_first = half == 0 ? t0 : t1, t_step = 1;
#else
      const int t_first = t0 + half, t_step = 2;
#endif
      if (t_first < t1) fetch_tile(t_first);
      for (int tile = t_first; tile < t1; tile += t_step) {    // the halves take alternate tiles of the workgroup's range
        const int first = lo + tile * Tsamp;
        const int32_t* samp = a.order + first;
        const int nS = min(Tsamp, hi - first), ntok = nS * F;
        const bool valid = tok < ntok;
        const int f = valid ? f_tok : 0;
        const int b = b_next;
        const size_t grow = ((size_t)b * F + f) * D + g4;       // this lane's row of x / dy / dx
        const int ntask = nS * H * F;
        const bool task_ok = tid_h < ntask;
        const int task = task_ok ? tid_h : 0;
        const int tls = task_ok ? t0_ls : 0, th = task_ok ? t0_h : 0, ti = task_ok ? t0_i : 0;

        // token-wise state that lives from phase A to phase F
        // (the MetaNet hidden rows are NOT kept: 32 registers for the whole tile against 2 x 32 MFMAs to rebuild them in phase F,
        // on a matrix pipe that idles two thirds of the time)
        float q0[KT][4], k0[KT][4], zhq[KT][4], zhk[KT][4], dr[KT][4];
        float rstd_q = 0.f, rstd_k = 0.f;
        // keep flags of this token lane at the MetaNet-Q / MetaNet-K / output sites (bits 0-7 / 8-15 / 16-23), generated once
        uint32_t keepbits = 0xFFFFFFFFu;
        if (dc.on)
            keepbits = token_keep_bits<KT>(drop_sample_key(dc.key[kSiteMetaQ], (uint32_t)b), f, D, g4, dc.thresh) |
                       (token_keep_bits<KT>(drop_sample_key(dc.key[kSiteMetaK], (uint32_t)b), f, D, g4, dc.thresh) << 8) |
                       (token_keep_bits<KT>(drop_sample_key(dc.key[kSiteOut], (uint32_t)b), f, D, g4, dc.thresh) << 16);

        STAMP(0);
        // ================= phase A: forward chain (every wave, also one whose rows lie beyond the tile: x = 0 there) =======
        {
            float x[KT][4], v[KT][4], q[KT][4], k[KT][4], hq[UT][4], hk[UT][4];
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[t][r] = x_next[t][r];
            chain<KT, KT, LD>(wq + lo_d, x, q0);
            chain<KT, KT, LD>(wk + lo_d, x, k0);
            chain<KT, KT, LD>(wv + lo_d, x, v);
            if (meta_q) {
                float m[KT][4];
                chain<KT, UT, LU>(w1q + lo_u, q0, hq);
#pragma unroll
                for (int t = 0; t < UT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hq[t][r] = fmaxf(hq[t][r], 0.f);
                chain<UT, KT, LD>(w2q + lo_d, hq, m);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float mm = m[t][r];
                        if (dc.on) mm = (keepbits >> (4 * t + r)) & 1u ? mm * dc.scale : 0.f;
                        m[t][r] = mm + q0[t][r];
                    }
                layer_norm_keep<KT>(m, zhq, rstd_q);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const float4 gg = *reinterpret_cast<const float4*>(lnq_g + 16 * t + g4);
                    const float4 bb = *reinterpret_cast<const float4*>(lnq_b + 16 * t + g4);
                    q[t][0] = zhq[t][0] * gg.x + bb.x; q[t][1] = zhq[t][1] * gg.y + bb.y;
                    q[t][2] = zhq[t][2] * gg.z + bb.z; q[t][3] = zhq[t][3] * gg.w + bb.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) q[t][r] = q0[t][r];
            }
            if (meta_k) {
                float m[KT][4];
                chain<KT, UT, LU>(w1k + lo_u, k0, hk);
#pragma unroll
                for (int t = 0; t < UT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hk[t][r] = fmaxf(hk[t][r], 0.f);
                chain<UT, KT, LD>(w2k + lo_d, hk, m);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float mm = m[t][r];
                        if (dc.on) mm = (keepbits >> (8 + 4 * t + r)) & 1u ? mm * dc.scale : 0.f;
                        m[t][r] = mm + k0[t][r];
                    }
                layer_norm_keep<KT>(m, zhk, rstd_k);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const float4 gg = *reinterpret_cast<const float4*>(lnk_g + 16 * t + g4);
                    const float4 bb = *reinterpret_cast<const float4*>(lnk_b + 16 * t + g4);
                    k[t][0] = zhk[t][0] * gg.x + bb.x; k[t][1] = zhk[t][1] * gg.y + bb.y;
                    k[t][2] = zhk[t][2] * gg.z + bb.z; k[t][3] = zhk[t][3] * gg.w + bb.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) k[t][r] = k0[t][r];
            }
            store_frag<KT>(sq + my, q);
            store_frag<KT>(sk + my, k);
            store_frag<KT>(sv + my, v);
        }
        half_barrier(hb_ctr, hb_round);
        STAMP(1);

        // ================= phase B: attention forward: o_i, row statistics, dropout keep word ===============================
        // two passes over the keys in chunks of four (rolled loops: the live set stays small next to the token state that
        // sleeps in registers): maximum of the scaled scores, then exp2 / sum / PV with the scores recomputed (4 packed FMAs
        // per key - cheaper than keeping the row in registers at two waves per SIMD); padding keys of the last chunk read the
        // last real row and are masked
        if (task_ok) {
            const int tb = samp[tls];
            f32x2 qi[d / 2];
            load_row<d>(sq + (tls * F + ti) * LD + th * d, qi);
            const float* kbase = sk + (tls * F) * LD + th * d;
            const float* vbase = sv + (tls * F) * LD + th * d;
            float mx = -INFINITY;
#pragma unroll 1
            for (int j0 = 0; j0 < F; j0 += 4) {
                f32x2 kr[4][d / 2];
#pragma unroll
                for (int u = 0; u < 4; ++u) load_row<d>(kbase + min(j0 + u, F - 1) * LD, kr[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) mx = fmaxf(mx, dot_row<d>(qi, kr[u]) * sc_scale);   // a padding key repeats the last score
            }
            f32x2 oacc[d / 2];
#pragma unroll
            for (int e = 0; e < d / 2; ++e) oacc[e] = f32x2{0.f, 0.f};
            float sum = 0.f;
            uint32_t keep = 0xFFFFFFFFu;
            const uint32_t skey = drop_sample_key(dc.key[kSiteAttn], (uint32_t)tb);
            const uint32_t block0 = drop_attn_elem(th, F, ti, 0) >> 2;
#pragma unroll 1
            for (int j0 = 0; j0 < F; j0 += 4) {
                f32x2 kr[4][d / 2], vr[4][d / 2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    load_row<d>(kbase + min(j0 + u, F - 1) * LD, kr[u]);
                    load_row<d>(vbase + min(j0 + u, F - 1) * LD, vr[u]);
                }
                const uint32_t kb = dc.on ? drop_keep4(skey, block0 + (uint32_t)(j0 >> 2), dc.thresh) : 0xFu;
                if (dc.on) keep = (keep & ~(0xFu << j0)) | (kb << j0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float ex = j0 + u < F ? __builtin_amdgcn_exp2f(dot_row<d>(qi, kr[u]) * sc_scale - mx) : 0.f;
                    sum += ex;
                    float pe = ex;
                    if (dc.on) pe = (kb >> u) & 1u ? ex * dc.scale : 0.f;
                    axpy_row<d>(pe, vr[u], oacc);
                }
            }
            const float inv = 1.0f / sum;
            st[task] = make_float4(mx, inv, 0.f, __uint_as_float(keep));
            store_row<d>(so + (tls * F + ti) * LD + th * d, oacc, inv);
        }
        half_barrier(hb_ctr, hb_round);
        STAMP(2);

        // ================= phase C: output block forward + backward ======================================================
        {
            float o[KT][4], u[KT][4], zh[KT][4], gy[KT][4], x[KT][4];
            load_frag<KT>(layer_x_row(a, b, f, F, D) + g4, x, valid);       // (L1 / L2 hits: phase A read these rows)
            load_frag<KT>(dy + grow, gy, valid);
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
            layer_norm_keep<KT>(u, zh, rstd_o);
            layer_norm_bwd_c<KT>(gy, zh, rstd_o, ln_g, g4, n, 0, 1, aln);       // gy is now dr
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dr[t][r] = use_res ? gy[t][r] : 0.f;
                    gy[t][r] *= keep[t][r];                                     // du
                }
            store_frag<KT>(x1 + my, gy);                                        // du rows (zero for padding tokens)
            float go[KT][4];
            chain_t<KT, KT, LD>(woT + lt_d, gy, go);                            // go = du Wo
            store_frag<KT>(x2 + my, go);
            // dot_i = sum_j P_ij dP_ij = go_i . o_i per head: features 16 t + 4 g + r belong to head 2 t + g / 2
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                float part = (go[t][0] * o[t][0] + go[t][1] * o[t][1]) + (go[t][2] * o[t][2] + go[t][3] * o[t][3]);
                part += __shfl_xor(part, 16, 64);
                if (valid && !(g & 1)) {
                    const int hh = 2 * t + (g >> 1);
                    reinterpret_cast<float*>(st + (ls_tok * H + hh) * F + f)[2] = part;
                }
            }
        }
        half_barrier(hb_ctr, hb_round);
        STAMP(3);

        // ================= phase D: dWo; softmax backward, row pass and column pass of every task =========================
        wgrad_split<KT, KT, KT, KT, LD>(x1, x1, so, so, wave, n, g, acc_wo);    // dWo[o][i] += du^T o
        f32x2 dq[d / 2], dk[d / 2], dv[d / 2];
#pragma unroll
        for (int e = 0; e < d / 2; ++e) { dq[e] = f32x2{0.f, 0.f}; dk[e] = f32x2{0.f, 0.f}; dv[e] = f32x2{0.f, 0.f}; }
        if (task_ok) {
            const float* qbase = sq + (tls * F) * LD + th * d;
            const float* kbase = sk + (tls * F) * LD + th * d;
            const float* vbase = sv + (tls * F) * LD + th * d;
            const float* gbase = x2 + (tls * F) * LD + th * d;
            const float4* stb = st + (tls * H + th) * F;
            const float scale = dc.scale;
            {   // row i: dq_i = sum_j dS_ij k_j,  dS_ij = P_ij (dP_ij - dot_i) / sqrt(d),  dP_ij = mask_ij (go_i . v_j)
                f32x2 qi[d / 2], gi[d / 2];
                load_row<d>(qbase + ti * LD, qi);
                load_row<d>(gbase + ti * LD, gi);
                const float4 s4 = stb[ti];
                const uint32_t keep = __float_as_uint(s4.w);
#pragma unroll 1
                for (int j0 = 0; j0 < F; j0 += 4) {
                    float dp[4];
                    {
                        f32x2 vr[4][d / 2];
#pragma unroll
                        for (int u = 0; u < 4; ++u) load_row<d>(vbase + min(j0 + u, F - 1) * LD, vr[u]);
#pragma unroll
                        for (int u = 0; u < 4; ++u) dp[u] = (keep >> (j0 + u)) & 1u ? dot_row<d>(gi, vr[u]) * scale : 0.f;
                    }
                    f32x2 kr[4][d / 2];
#pragma unroll
                    for (int u = 0; u < 4; ++u) load_row<d>(kbase + min(j0 + u, F - 1) * LD, kr[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float s_ = dot_row<d>(qi, kr[u]) * sc_scale;
                        const float pj = j0 + u < F ? __builtin_amdgcn_exp2f(s_ - s4.x) * s4.y : 0.f;
                        axpy_row<d>(pj * (dp[u] - s4.z) * inv_sqrt_d, kr[u], dq);
                    }
                }
            }
            {   // column i: dk_i = sum_r dS_ri q_r,  dv_i = sum_r P_ri mask_ri go_r
                f32x2 ki[d / 2], vi[d / 2];
                load_row<d>(kbase + ti * LD, ki);
                load_row<d>(vbase + ti * LD, vi);
#pragma unroll 1
                for (int r0 = 0; r0 < F; r0 += 2) {
                    f32x2 qr[2][d / 2], gr[2][d / 2];
                    float4 sr[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int r = min(r0 + u, F - 1);
                        sr[u] = stb[r];
                        load_row<d>(qbase + r * LD, qr[u]);
                        load_row<d>(gbase + r * LD, gr[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const bool real = r0 + u < F;
                        const float s_ = dot_row<d>(qr[u], ki) * sc_scale;
                        const float pr = real ? __builtin_amdgcn_exp2f(s_ - sr[u].x) * sr[u].y : 0.f;
                        const bool kp = (__float_as_uint(sr[u].w) >> ti) & 1u;
                        const float dp = kp ? dot_row<d>(gr[u], vi) * scale : 0.f;
                        axpy_row<d>(pr * (dp - sr[u].z) * inv_sqrt_d, qr[u], dk);
                        axpy_row<d>(kp ? pr * scale : 0.f, gr[u], dv);
                    }
                }
            }
        }
        half_barrier(hb_ctr, hb_round);
        STAMP(4);
        if (task_ok) {      // every q, k, v, go row has been read (and du, o by the dWo product)
            store_row<d>(x1 + (tls * F + ti) * LD + th * d, dq, 1.0f);
            store_row<d>(sk + (tls * F + ti) * LD + th * d, dk, 1.0f);
            store_row<d>(sv + (tls * F + ti) * LD + th * d, dv, 1.0f);
        }
        half_barrier(hb_ctr, hb_round);
        STAMP(5);

        // ================= phase F: MetaNet and projection backward, dx; split weight gradients in three rounds ============
        // Exchange buffers: E0 = sq, E1 = so, E2 = x2, E3 = x1 (after gq is loaded), E4 = sk (after gk is loaded), sv after dWv.
        // A wave only ever writes its OWN 16 rows; the split products read all 128 rows, hence the barriers.
        {
            float gq[KT][4], gk[KT][4], x[KT][4];
            load_frag<KT>(layer_x_row(a, b, f, F, D) + g4, x, valid);
            load_frag<KT>(x1 + my, gq, valid);      // gradient of the (post-MetaNet) queries
            load_frag<KT>(sk + my, gk, valid);      // ... keys
            {   // dx so far: dr + gv Wv^T   (rows >= ntok of sv still hold forward values: masked here, and neutralised by
                // x = 0 in the dWv product)
                float gv[KT][4], back[KT][4];
                load_frag<KT>(sv + my, gv, valid);
                chain_t<KT, KT, LD>(wv + lt_d, gv, back);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dr[t][r] += back[t][r];
            }
            // MetaNet backward of one role up to the point where its two exchange rounds start: LN backward (gout becomes
            // dz), dm = dz * mask, dh = (dm W2^T) * [h > 0], gout = dz + dh W1^T
            auto metanet_bwd = [&](float (&gout)[KT][4], const float (&zh)[KT][4], float rstd, const float* gam, int vg,
                                   int kshift, const float (&in0)[KT][4], float (&h)[UT][4], const float* w2, const float* w1,
                                   float (&dm)[KT][4], float (&dh)[UT][4]) {
                chain<KT, UT, LU>(w1 + lo_u, in0, h);                        // the hidden rows again: relu(in0 W1)
#pragma unroll
                for (int t = 0; t < UT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[t][r] = fmaxf(h[t][r], 0.f);
                layer_norm_bwd_c<KT>(gout, zh, rstd, gam, g4, n, vg, vg + 1, aln);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float mk = 1.0f;
                        if (dc.on) mk = (keepbits >> (kshift + 4 * t + r)) & 1u ? dc.scale : 0.f;
                        dm[t][r] = gout[t][r] * mk;
                    }
                chain_t<KT, UT, LD>(w2 + lt_d, dm, dh);                      // w2: forward image W2 [U][LD]
#pragma unroll
                for (int t = 0; t < UT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dh[t][r] = h[t][r] > 0.f ? dh[t][r] : 0.f;
                float back[KT][4];
                chain_t<UT, KT, LU>(w1 + lt_u, dh, back);                    // w1: forward image W1 [D][LU]
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gout[t][r] += back[t][r];
            };
            auto store_wide = [&](float* b0, float* b1, const float (&w)[UT][4]) {
                float part[KT][4];
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[t][r] = w[t][r];
                store_frag<KT>(b0 + my, part);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[t][r] = w[KT + t][r];
                store_frag<KT>(b1 + my, part);
            };

            // ---- round 1: {x, dv} -> dWv ; {hq, dmq} -> dW2 (Q role) -------------------------------------------------------
            float dhq[UT][4];
            store_frag<KT>(sq + my, x);
            if (meta_q) {
                float dm[KT][4], hq[UT][4];
                metanet_bwd(gq, zhq, rstd_q, lnq_g, 2, 0, q0, hq, w2q, w1q, dm, dhq);
                store_wide(so, x2, hq);
                store_frag<KT>(x1 + my, dm);
            }
            half_barrier(hb_ctr, hb_round);
            STAMP(6);
            wgrad_split<KT, KT, KT, KT, LD>(sq, sq, sv, sv, wave, n, g, acc_wv);                 // dWv[i][o] += x^T dv
            if (meta_q) wgrad_split<UT, KT, KT, KT, LD>(so, x2, x1, x1, wave, n, g, acc_w2q);   // dW2[u][o] += hq^T dm
            half_barrier(hb_ctr, hb_round);
            STAMP(7);
            // ---- round 2: {q0, dhq} -> dW1 (Q role) ; {hk, dmk} -> dW2 (K role) ---------------------------------------------
            float dhk[UT][4];
            if (meta_q) {
                store_frag<KT>(sq + my, q0, valid);
                store_wide(so, x2, dhq);
            }
            if (meta_k) {
                float dm[KT][4], hk[UT][4];
                metanet_bwd(gk, zhk, rstd_k, lnk_g, 4, 8, k0, hk, w2k, w1k, dm, dhk);
                store_wide(x1, sv, hk);
                store_frag<KT>(sk + my, dm);
            }
            half_barrier(hb_ctr, hb_round);
            STAMP(8);
            if (meta_q) wgrad_split<KT, UT, KT, KT, LD>(sq, sq, so, x2, wave, n, g, acc_w1q);   // dW1[i][u] += q0^T dhq
            if (meta_k) {
                if constexpr (SAME) wgrad_split<UT, KT, KT, KT, LD>(x1, sv, sk, sk, wave, n, g, acc_w2q);
                else wgrad_split<UT, KT, KT, KT, LD>(x1, sv, sk, sk, wave, n, g, acc_w2k);
            }
            half_barrier(hb_ctr, hb_round);
            STAMP(9);
            // ---- round 3: {k0, dhk} -> dW1 (K role) ; {x, gq, gk} -> dWq, dWk -------------------------------------------------
            if (meta_k) {
                store_frag<KT>(sq + my, k0, valid);
                store_wide(so, x2, dhk);
            }
            store_frag<KT>(x1 + my, x);
            store_frag<KT>(sv + my, gq);
            store_frag<KT>(sk + my, gk);
            if (tile + t_step < t1) fetch_tile(tile + t_step);
            half_barrier(hb_ctr, hb_round);
            STAMP(10);
            if (meta_k) {
                if constexpr (SAME) wgrad_split<KT, UT, KT, KT, LD>(sq, sq, so, x2, wave, n, g, acc_w1q);
                else wgrad_split<KT, UT, KT, KT, LD>(sq, sq, so, x2, wave, n, g, acc_w1k);
            }
            wgrad_split<KT, KT, KT, KT, LD>(x1, x1, sv, sv, wave, n, g, acc_wq);                 // dWq[i][o] += x^T gq
            wgrad_split<KT, KT, KT, KT, LD>(x1, x1, sk, sk, wave, n, g, acc_wk);
            // dx = dr + gq Wq^T + gk Wk^T (+ gv Wv^T above)
            float back[KT][4];
            chain_t<KT, KT, LD>(wq + lt_d, gq, back);
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dr[t][r] += back[t][r];
            chain_t<KT, KT, LD>(wk + lt_d, gk, back);
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dr[t][r] += back[t][r];
            if (valid) store_frag<KT>(dx + grow, dr);
        }
        half_barrier(hb_ctr, hb_round);
        STAMP(11);
      }
      __syncthreads();      // both halves are through with this scenario's tiles (and with the row buffers the flush stages in)
      // ---- this scenario's generated-weight gradients: record (workgroup + scenario) ---------------------------------
      // with one shared table both roles accumulated into acc_w1q / acc_w2q; the reducer reads the part of a role only
      // when that role is active, so the sums go to the Q part when Q is modulated, else to the K part
      {
          float* rec = records + (size_t)(blockIdx.x + scen) * TSZ;
          const bool to_k = SAME && !meta_q;
          flush(acc_w1q, KTc{}, UTc{}, rec + (to_k ? 2 * D * U : 0), true);
          flush(acc_w2q, UTc{}, KTc{}, rec + (to_k ? 3 * D * U : D * U), true);
          flush(acc_w1k, KTc{}, UTc{}, rec + (to_k ? 0 : 2 * D * U), !SAME);
          flush(acc_w2k, UTc{}, KTc{}, rec + (to_k ? D * U : 3 * D * U), !SAME);
      }
    }

    STAMP(12);
    // ---- scenario-independent gradients of this workgroup ------------------------------------------------------------------
    flush(acc_wq, KTc{}, KTc{}, common, true);
    flush(acc_wk, KTc{}, KTc{}, common + D * D, true);
    flush(acc_wv, KTc{}, KTc{}, common + 2 * D * D, true);
    flush(acc_wo, KTc{}, KTc{}, common + 3 * D * D, true);
    // LayerNorm gradients: lane n < 6 of every 16-lane row holds vector n (features 16 t + 4 g + r); waves in wave order
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n < 6) stage[(wave8 * 6 + n) * D + 16 * t + g4 + r] = aln[t][r];
    __syncthreads();
    for (int e = threadIdx.x; e < 6 * D; e += kB8Block) {
        float s = stage[e];
#pragma unroll
        for (int w = 1; w < kB8Waves; ++w) s += stage[w * 6 * D + e];
        common[4 * D * D + e] = s;
    }
}

static int64_t bwd8_lds_floats(int T, int F, int D, int U, int H, bool same_tab) {
    const int LD = D + 4, LU = U + 4;
    auto r4 = [](int64_t v) { return (v + 3) & ~(int64_t)3; };
    return 4 * (int64_t)D * LD + (same_tab ? 1 : 2) * ((int64_t)D * LU + (int64_t)U * LD) + 6 * D + 2 * 6 * (int64_t)kB8Rows * LD +
           r4(2 * 4 * (int64_t)T * H * F) + 64;
}

struct Bwd8Plan {
    int T, G, FC;
    size_t lds;
};

// -1: not decided yet (SATRANS_BWD8 in the environment, default 0); 0 / 1: off / on (satrans_set_layer_bwd8)
static int g_bwd8 = -1;

static bool bwd8_plan(const satrans_layer_desc* d, Bwd8Plan& p) {
    if (g_bwd8 < 0) g_bwd8 = getenv("SATRANS_BWD8") ? (atoi(getenv("SATRANS_BWD8")) != 0) : 0;
    if (!g_bwd8) return false;
    if (d->flags & (SATRANS_GATE | SATRANS_BILINEAR)) return false;
    const bool meta = d->flags & (SATRANS_META_Q | SATRANS_META_K);
    const bool shape = (d->D == 32 && d->H == 4 && (!meta || d->U == 64)) || (d->D == 16 && d->H == 2 && (!meta || d->U == 32));
    if (!shape || d->F > 32 || d->F < 1) return false;
    const bool same_tab = d->tab_q == d->tab_k;
    const int U = 2 * d->D;
    p.T = kB8Rows / d->F;
    static const int force_t = getenv("SATRANS_BWD8_T") ? atoi(getenv("SATRANS_BWD8_T")) : 0;   // (experiments)
    if (force_t > 0 && force_t < p.T) p.T = force_t;
    p.FC = d->F <= 20 ? 5 : 8;
    p.lds = (size_t)bwd8_lds_floats(p.T, d->F, d->D, U, d->H, same_tab) * 4;
    if (p.lds > 160 * 1024) return false;
    if (d->D == 32 && !same_tab) return false;                               // (not instantiated: does not fit LDS)
    const int64_t tiles = ceil_div(d->B, p.T) + d->S;
    p.G = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, cu_count()));      // one workgroup per CU, one round
    return true;
}

template <int D, int U, int H, bool SAME, int FC, int FT>
static int launch_bwd8(const satrans_layer_desc* d, const Bwd8Plan& p, const float* dy, float* dx, float* slabs,
                       hipStream_t stream) {
    static size_t attr_set = 0;
    if (p.lds > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)layer_bwd8_kernel<D, U, H, SAME, FC, FT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "layer_bwd8: LDS attribute: %s", hipGetErrorString(e));
        attr_set = p.lds;
    }
    layer_bwd8_kernel<D, U, H, SAME, FC, FT><<<p.G, kB8Block, p.lds, stream>>>(*d, p.T, dy, dx, slabs);
    SATRANS_CHECK_LAUNCH("layer_bwd8_kernel");
    return SATRANS_OK;
}

}  // namespace satrans

using namespace satrans;

#ifdef SATRANS_STAMPS
extern "C" int satrans_debug_read_stamps8(unsigned long long* h_out, int reset) {
    if (hipMemcpyFromSymbol(h_out, HIP_SYMBOL(satrans::g_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(satrans::g_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" int satrans_set_layer_bwd8(int on) {
    satrans::g_bwd8 = on < 0 ? -1 : (on != 0);
    return SATRANS_OK;
}

extern "C" int satrans_layer_bwd8_supported(const satrans_layer_desc* d) {
    Bwd8Plan p;
    return d && bwd8_plan(d, p) ? 1 : 0;
}

extern "C" int64_t satrans_layer_bwd8_slab_floats(const satrans_layer_desc* d) {
    Bwd8Plan p;
    if (!d || !bwd8_plan(d, p)) return -1;
    const int64_t U = 2 * (int64_t)d->D;
    const int64_t CSZ = 4 * (int64_t)d->D * d->D + 6 * d->D, TSZ = 4 * (int64_t)d->D * U;
    return (int64_t)p.G * CSZ + (int64_t)(p.G + d->S) * TSZ;
}

// launches the kernel; *T_out / *G_out: the tile size and grid the fixed-order reduction must use
extern "C" int satrans_layer_bwd8_launch(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, int* T_out,
                                         int* G_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    Bwd8Plan p;
    SATRANS_REQUIRE(bwd8_plan(d, p), SATRANS_E_UNSUPPORTED, "layer_bwd8: shape not built");
    const bool same = d->tab_q == d->tab_k;
    int rc;
    if (d->D == 32 && d->F == 19) rc = launch_bwd8<32, 64, 4, true, 5, 19>(d, p, dy, dx, slabs, stream);   // the AliCCP shape
    else if (d->D == 32) rc = p.FC == 5 ? launch_bwd8<32, 64, 4, true, 5, 0>(d, p, dy, dx, slabs, stream)
                                        : launch_bwd8<32, 64, 4, true, 8, 0>(d, p, dy, dx, slabs, stream);
    else if (same) rc = p.FC == 5 ? launch_bwd8<16, 32, 2, true, 5, 0>(d, p, dy, dx, slabs, stream)
                                  : launch_bwd8<16, 32, 2, true, 8, 0>(d, p, dy, dx, slabs, stream);
    else rc = p.FC == 5 ? launch_bwd8<16, 32, 2, false, 5, 0>(d, p, dy, dx, slabs, stream)
                        : launch_bwd8<16, 32, 2, false, 8, 0>(d, p, dy, dx, slabs, stream);
    *T_out = p.T;
    *G_out = p.G;
    return rc;
}
