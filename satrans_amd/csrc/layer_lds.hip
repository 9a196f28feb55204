// Meta_Transformer_Layer forward and backward, shape-generic LDS version ("v1").
//
// Reference: models/satrans.py:50-100 (layer) and models/submodules.py:77-103 (MetaNet).
//
// Work decomposition.  The generated MetaNet weights depend only on the scenario id, so
// samples are bucketed by scenario (satrans_bucket_scenarios) and a workgroup only ever works on samples of
// ONE scenario: blockIdx.y = scenario row, blockIdx.x strides over that scenario's tiles of T samples.  All
// activations of a tile live in LDS ([token][feature], row stride padded by one float so that column walks
// are bank-conflict free); weights are staged into LDS once per use.  The backward pass recomputes the
// forward from the layer input with the same counter-based dropout masks (rng.h), so the only tensor kept
// between passes is the layer input.  Weight gradients are accumulated per workgroup into a private slab
// (fixed order over tiles) and summed over workgroups in index order by a second kernel: results are
// bitwise reproducible, no float atomics.
//
// Two flavours of the dense products share this control flow (LayerLds::mfma, chosen on the host):
//   * MFMA: v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate: bit-for-bit a k-ordered fmaf chain, so the fp32 parity
//     bar is kept) with the TOKENS on the N side (lane & 15) and the output features on the M side, activations
//     read from the [token][feature] LDS images as 16-byte B fragments, weights as conflict-free 4-byte A
//     fragments (row stride padded by 4 floats).  Needs D % 16 == 0 and U % 16 == 0.
//   * scalar FMA loops: any F, any D multiple of 4, any H | D, any U (the fallback and the ablation arm).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "rng.h"

extern "C" int satrans_layer_impl(void);

namespace satrans {

constexpr int kLayerBlock = 256;

struct LayerLds {
    // forward activations, [tok][ldd] unless noted
    float *x, *q0, *k0, *v;
    float *hq, *hk;        // [tok][ldu] MetaNet hidden (post-ReLU)
    float *zq, *q, *zk, *k;  // pre-/post-LayerNorm MetaNet outputs
    float* P;              // [nS*H*F][ldp] softmax probabilities BEFORE dropout
    float *o, *u, *r;      // attention output, Out_linear output (kept only when non-null), pre-LayerNorm residual sum
    float* w;              // weight staging
    int ldd, ldu, ldp;
    int mfma;              // 1: MFMA products, 0: scalar FMA loops
    int wp;                // row padding (floats) of staged weight matrices: 4 with MFMA, else 0
};

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct TileDims {
    int F, D, H, U, d;
    int nS, ntok;
};

// MFMA form of out(t, o) = sum_k in[t][k] * w[k][o].  One work item = 16 tokens x 16 output features, one item per
// wave at a time.  v_mfma_f32_16x16x4_f32 operand maps (cdna_hip_programming.md §3): A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15], D[row = 4*(lane>>4) + reg][col = lane&15].  Tokens sit on j, output features on i.
// The contraction index of MFMA step (t, r) held by lane group g is feature 16t + 4g + r, so a lane's four B
// values of one t are ONE 16-byte LDS read of its token row, and the accumulator comes out as four consecutive
// output features of one token.  Rows >= ntok of `in` may hold anything (a B column only feeds its own D column).
template <typename Epi>
__device__ __forceinline__ void gemm_tokens_mfma(const float* __restrict__ in, int ldin, const float* __restrict__ w,
                                                 int ldw, int K, int N, int ntok, Epi epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int ntt = (ntok + 15) >> 4, nob = N >> 4;
    for (int item = wave; item < ntt * nob; item += nw) {
        const int tt = item / nob, ot = item - tt * nob;
        const float* brow = in + (size_t)(16 * tt + n) * ldin + 4 * g;
        const float* acol = w + (size_t)(4 * g) * ldw + 16 * ot + n;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < (K >> 4); ++t) {
            const float4 b = *reinterpret_cast<const float4*>(brow + 16 * t);
            const float* ac = acol + (size_t)(16 * t) * ldw;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[0], b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[ldw], b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[2 * ldw], b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[3 * ldw], b.w, acc, 0, 0, 0);
        }
        const int tok = 16 * tt + n, o0 = 16 * ot + 4 * g;
        if (tok < ntok) {
            epi(tok, o0, acc[0]);
            epi(tok, o0 + 1, acc[1]);
            epi(tok, o0 + 2, acc[2]);
            epi(tok, o0 + 3, acc[3]);
        }
    }
}

// out(t, o) = sum_k in[t][k] * w[k][o] for t < ntok, o < N; `epi(t, o, acc)` consumes the result.
// Scalar form: one work item = TR tokens x 1 output column; consecutive lanes take consecutive columns, so the
// weight reads are conflict-free and the activation reads are LDS broadcasts.
template <int TR, typename Epi>
__device__ __forceinline__ void gemm_tokens(bool mfma, const float* __restrict__ in, int ldin,
                                            const float* __restrict__ w, int ldw, int K, int N, int ntok, Epi epi) {
    if (mfma) {
        gemm_tokens_mfma(in, ldin, w, ldw, K, N, ntok, epi);
        return;
    }
    const int groups = (ntok + TR - 1) / TR;
    for (int item = threadIdx.x; item < groups * N; item += blockDim.x) {
        const int tg = item / N, o = item - tg * N;
        const int t0 = tg * TR;
        float acc[TR];
        const float* row[TR];
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            acc[r] = 0.f;
            row[r] = in + (size_t)min(t0 + r, ntok - 1) * ldin;
        }
        for (int k = 0; k < K; ++k) {
            const float wv = w[k * ldw + o];
#pragma unroll
            for (int r = 0; r < TR; ++r) acc[r] = fmaf(row[r][k], wv, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < TR; ++r)
            if (t0 + r < ntok) epi(t0 + r, o, acc[r]);
    }
}

// s[r*ld + c] = g[r*C + c]   (g is [R][C] row-major)
__device__ __forceinline__ void stage_rows(const float* __restrict__ g, float* __restrict__ s, int R, int C, int ld) {
    for (int i = threadIdx.x; i < R * C; i += blockDim.x) {
        const int r = i / C, c = i - r * C;
        s[r * ld + c] = g[i];
    }
}

// s[c*ld + r] = g[r*C + c]   (g is [R][C] row-major; s becomes [C][R] with row stride ld)
__device__ __forceinline__ void stage_transposed(const float* __restrict__ g, float* __restrict__ s, int R, int C,
                                                 int ld) {
    for (int i = threadIdx.x; i < R * C; i += blockDim.x) {
        const int r = i / C, c = i - r * C;
        s[c * ld + r] = g[i];
    }
}

// LayerNorm statistics of one row (torch: biased variance, eps inside the sqrt)
__device__ __forceinline__ void row_stats(const float* row, int D, float& mean, float& rstd) {
    float s = 0.f;
    for (int c = 0; c < D; ++c) s += row[c];
    mean = s / (float)D;
    float v = 0.f;
    for (int c = 0; c < D; ++c) {
        const float e = row[c] - mean;
        v = fmaf(e, e, v);
    }
    rstd = 1.0f / sqrtf(v / (float)D + 1e-6f);
}

__device__ __forceinline__ void layer_norm_rows(const float* in, float* out, int ld, int ntok, int D,
                                                const float* __restrict__ gamma, const float* __restrict__ beta) {
    for (int t = threadIdx.x; t < ntok; t += blockDim.x) {
        float mean, rstd;
        row_stats(in + (size_t)t * ld, D, mean, rstd);
        for (int c = 0; c < D; ++c) out[(size_t)t * ld + c] = (in[(size_t)t * ld + c] - mean) * rstd * gamma[c] + beta[c];
    }
}

struct DropCtx {
    bool on;
    float scale;
    uint32_t thresh;
    uint32_t key[4];  // per site
};

__device__ __forceinline__ DropCtx make_drop(const satrans_layer_desc& a) {
    DropCtx dc;
    dc.on = (a.flags & SATRANS_TRAIN) && a.drop_p > 0.f;
    dc.scale = dc.on ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    dc.thresh = drop_threshold(a.drop_p);
    for (int s = 0; s < 4; ++s) dc.key[s] = drop_site_key(a.seed, a.step, a.layer, s);
    return dc;
}

// multiplicative mask value (0 or 1/(1-p)) of element `elem` of global sample `b` at `site`
__device__ __forceinline__ float drop_mask(const DropCtx& dc, int site, int b, uint32_t elem) {
    if (!dc.on) return 1.0f;
    return drop_keep(drop_sample_key(dc.key[site], (uint32_t)b), elem, dc.thresh) ? dc.scale : 0.0f;
}

// MetaNet of one role (Q or K): z = drop(relu(in @ W1) @ W2) + in ; out = LN(z).   submodules.py:77-103
__device__ void metanet_tile(const satrans_layer_desc& a, const TileDims& T, const int32_t* samp, const DropCtx& dc,
                             int site, const float* __restrict__ tab_row, const float* gamma, const float* beta,
                             const float* in, float* h, float* z, float* out, LayerLds& L) {
    const int D = T.D, U = T.U, F = T.F;
    stage_rows(tab_row, L.w, D, U, U + L.wp);  // W1 [D][U]
    __syncthreads();
    gemm_tokens<4>(L.mfma, in, L.ldd, L.w, U + L.wp, D, U, T.ntok,
                   [&](int t, int o, float acc) { h[(size_t)t * L.ldu + o] = fmaxf(acc, 0.f); });
    __syncthreads();
    stage_rows(tab_row + D * U, L.w, U, D, D + L.wp);  // W2 [U][D]
    __syncthreads();
    gemm_tokens<4>(L.mfma, h, L.ldu, L.w, D + L.wp, U, D, T.ntok, [&](int t, int o, float acc) {
        const int ls = t / F, f = t - ls * F;
        const float m = acc * drop_mask(dc, site, samp[ls], (uint32_t)(f * D + o));
        z[(size_t)t * L.ldd + o] = m + in[(size_t)t * L.ldd + o];
    });
    __syncthreads();
    layer_norm_rows(z, out, L.ldd, T.ntok, D, gamma, beta);
    __syncthreads();
}

// Forward of one tile up to the pre-LayerNorm residual sum r.  Buffers of L may alias as documented at the
// call sites (forward kernel) or be distinct (backward kernel).
__device__ void forward_tile(const satrans_layer_desc& a, const TileDims& T, int scen, const int32_t* samp,
                             const DropCtx& dc, LayerLds& L) {
    const int F = T.F, D = T.D, H = T.H, d = T.d, ntok = T.ntok;
    // ---- load x, stage [Wq | Wk | Wv] as [D][3D] -------------------------------------------------------
    for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
        const int t = i / D, c = i - t * D;
        const int ls = t / F, f = t - ls * F;
        L.x[(size_t)t * L.ldd + c] = layer_x_row(a, samp[ls], f, F, D)[c];
    }
    const int ld3 = 3 * D + L.wp;
    for (int i = threadIdx.x; i < D * D; i += blockDim.x) {
        const int k = i / D, o = i - k * D;
        L.w[k * ld3 + o] = a.w_query[i];
        L.w[k * ld3 + D + o] = a.w_key[i];
        L.w[k * ld3 + 2 * D + o] = a.w_value[i];
    }
    __syncthreads();
    gemm_tokens<4>(L.mfma, L.x, L.ldd, L.w, ld3, D, 3 * D, ntok, [&](int t, int o, float acc) {   // satrans.py:55-57
        if (o < D) L.q0[(size_t)t * L.ldd + o] = acc;
        else if (o < 2 * D) L.k0[(size_t)t * L.ldd + o - D] = acc;
        else L.v[(size_t)t * L.ldd + o - 2 * D] = acc;
    });
    __syncthreads();
    // ---- scenario modulation of Q and K (satrans.py:60-81) ----------------------------------------------------
    auto copy_rows = [&](const float* src, float* dst) {
        for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
            const int t = i / D, c = i - t * D;
            dst[(size_t)t * L.ldd + c] = src[(size_t)t * L.ldd + c];
        }
    };
    if (a.flags & SATRANS_GATE) {
        // flag 'gate' (satrans.py:61-62,68-69): q = q * vec * 2 with the generated row of length D
        const float* gq = a.tab_q + (size_t)scen * a.tab_stride;
        const float* gk = a.tab_k + (size_t)scen * a.tab_stride;
        for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
            const int t = i / D, c = i - t * D;
            const size_t at = (size_t)t * L.ldd + c;
            L.q[at] = (a.flags & SATRANS_META_Q) ? L.q0[at] * gq[c] * 2.0f : L.q0[at];
            L.k[at] = (a.flags & SATRANS_META_K) ? L.k0[at] * gk[c] * 2.0f : L.k0[at];
        }
        __syncthreads();
    } else if (a.flags & SATRANS_BILINEAR) {
        // flag 'bilinear' (satrans.py:79-81): per head q_h <- q_h @ M[h], M = generated row viewed as [H][d][d].
        // Not in place: the products go to the hidden buffer first.
        const float* M = a.tab_q + (size_t)scen * a.tab_stride;
        for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
            const int t = i / D, c = i - t * D;
            const int h = c / d, e2 = c - h * d;
            const float* qrow = L.q0 + (size_t)t * L.ldd + h * d;
            float acc = 0.f;
            for (int e = 0; e < d; ++e) acc = fmaf(qrow[e], M[(h * d + e) * d + e2], acc);
            L.hq[(size_t)t * L.ldd + c] = acc;
        }
        __syncthreads();
        copy_rows(L.hq, L.q);
        if (L.k != L.k0) copy_rows(L.k0, L.k);
        __syncthreads();
    } else {
        // MetaNet on Q and K (satrans.py:60-73)
        if (a.flags & SATRANS_META_Q) {
            metanet_tile(a, T, samp, dc, kSiteMetaQ, a.tab_q + (size_t)scen * a.tab_stride, a.lnq_g, a.lnq_b, L.q0,
                         L.hq, L.zq, L.q, L);
        } else if (L.q != L.q0) {
            copy_rows(L.q0, L.q);
            __syncthreads();
        }
        if (a.flags & SATRANS_META_K) {
            metanet_tile(a, T, samp, dc, kSiteMetaK, a.tab_k + (size_t)scen * a.tab_stride, a.lnk_g, a.lnk_b, L.k0,
                         L.hk, L.zk, L.k, L);
        } else if (L.k != L.k0) {
            copy_rows(L.k0, L.k);
            __syncthreads();
        }
    }
    // ---- scores + softmax: one thread per (sample, head, query row)  (satrans.py:84-87) --------------------
    const float sqrt_d = sqrtf((float)d);
    for (int task = threadIdx.x; task < T.nS * H * F; task += blockDim.x) {
        const int ls = task / (H * F), rem = task - ls * H * F;
        const int h = rem / F, i = rem - h * F;
        const float* qi = L.q + (size_t)(ls * F + i) * L.ldd + h * d;
        float* prow = L.P + (size_t)task * L.ldp;
        float mx = -INFINITY;
        for (int j = 0; j < F; ++j) {
            const float* kj = L.k + (size_t)(ls * F + j) * L.ldd + h * d;
            float s = 0.f;
            for (int e = 0; e < d; ++e) s = fmaf(qi[e], kj[e], s);
            s = s / sqrt_d;
            prow[j] = s;
            mx = fmaxf(mx, s);
        }
        float sum = 0.f;
        for (int j = 0; j < F; ++j) {
            const float e = expf(prow[j] - mx);
            prow[j] = e;
            sum += e;
        }
        for (int j = 0; j < F; ++j) prow[j] = prow[j] / sum;
    }
    __syncthreads();
    // ---- o = dropout(P) @ V, heads written back side by side (satrans.py:88-90) ------------------------------
    for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
        const int t = i / D, c = i - t * D;
        const int ls = t / F, qi = t - ls * F, h = c / d;
        const int task = (ls * H + h) * F + qi;
        const float* prow = L.P + (size_t)task * L.ldp;
        const int b = samp[ls];
        float acc = 0.f;
        for (int j = 0; j < F; ++j) {
            const float p = prow[j] * drop_mask(dc, kSiteAttn, b, drop_attn_elem(h, F, qi, j));
            acc = fmaf(p, L.v[(size_t)(ls * F + j) * L.ldd + c], acc);
        }
        L.o[(size_t)t * L.ldd + c] = acc;
    }
    stage_transposed(a.w_out, L.w, D, D, D + L.wp);  // w[k][o] = Wo[o][k]   (nn.Linear: y = x @ Wo^T)
    __syncthreads();
    // ---- r = dropout(act(o @ Wo^T)) + x  (satrans.py:91-97) ---------------------------------------------------
    gemm_tokens<4>(L.mfma, L.o, L.ldd, L.w, D + L.wp, D, D, ntok, [&](int t, int o, float acc) {
        const int ls = t / F, f = t - ls * F;
        if (L.u) L.u[(size_t)t * L.ldd + o] = acc;
        if (a.flags & SATRANS_RELU_OUT) acc = fmaxf(acc, 0.f);
        acc *= drop_mask(dc, kSiteOut, samp[ls], (uint32_t)(f * D + o));
        if (!(a.flags & SATRANS_NO_RES)) acc += L.x[(size_t)t * L.ldd + o];
        L.r[(size_t)t * L.ldd + o] = acc;
    });
    __syncthreads();
}

__device__ __forceinline__ bool tile_of(const satrans_layer_desc& a, int scen, int tile, int Tsamp, const int32_t*& samp,
                                        int& nS) {
    const int lo = a.seg[scen], hi = a.seg[scen + 1];
    const int first = lo + tile * Tsamp;
    if (first >= hi) return false;
    samp = a.order + first;
    nS = min(Tsamp, hi - first);
    return true;
}

// ------------------------------------------------------------------------------------------------------
// Forward kernel.  LDS aliasing: zq = q = q0 (MetaNet is applied in place), zk = k = k0, hk = hq,
// P shares storage with hq (dead before the scores are written), o = q (every (sample,head,row) task reads
// its own q row in the score phase, which ends at a barrier before o is written), r = k0.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kLayerBlock) void layer_fwd_kernel(satrans_layer_desc a, int Tsamp, int use_mfma,
                                                              float* __restrict__ y, float* __restrict__ att) {
    extern __shared__ __align__(16) float lds[];
    const int F = a.F, D = a.D, H = a.H, U = a.U;
    const int scen = blockIdx.y;
    TileDims T{F, D, H, U, D / H, 0, 0};
    LayerLds L;
    L.mfma = use_mfma;
    L.wp = use_mfma ? 4 : 0;
    L.ldd = D + (use_mfma ? 4 : 1);
    L.ldu = U + (use_mfma ? 4 : 1);
    L.ldp = F + 1;
    const int maxtok = Tsamp * F;
    float* p = lds;
    auto take = [&](int n) { float* r = p; p += (n + 3) & ~3; return r; };
    L.x = take(maxtok * L.ldd);
    L.q0 = take(maxtok * L.ldd);
    L.k0 = take(maxtok * L.ldd);
    L.v = take(maxtok * L.ldd);
    L.hq = L.hk = L.P = take(max(max(maxtok * L.ldu, maxtok * L.ldd), Tsamp * H * F * L.ldp));
    L.w = p;
    L.zq = L.q = L.q0;
    L.zk = L.k = L.k0;
    L.o = L.q0;
    L.u = nullptr;
    L.r = L.k0;
    const DropCtx dc = make_drop(a);

    for (int tile = blockIdx.x;; tile += gridDim.x) {
        const int32_t* samp;
        if (!tile_of(a, scen, tile, Tsamp, samp, T.nS)) break;
        T.ntok = T.nS * F;
        forward_tile(a, T, scen, samp, dc, L);
        if (att) {  // normalized_att_scores [H,B,F,F], after dropout (satrans.py:87)
            for (int i = threadIdx.x; i < T.nS * H * F * F; i += blockDim.x) {
                const int task = i / F, j = i - task * F;
                const int ls = task / (H * F), rem = task - ls * H * F;
                const int h = rem / F, qi = rem - h * F;
                const int b = samp[ls];
                const float pv = L.P[(size_t)task * L.ldp + j] * drop_mask(dc, kSiteAttn, b, drop_attn_elem(h, F, qi, j));
                att[(((size_t)h * a.B + b) * F + qi) * F + j] = pv;
            }
        }
        layer_norm_rows(L.r, L.r, L.ldd, T.ntok, D, a.ln_g, a.ln_b);  // satrans.py:99
        __syncthreads();
        for (int i = threadIdx.x; i < T.ntok * D; i += blockDim.x) {
            const int t = i / D, c = i - t * D;
            const int ls = t / F, f = t - ls * F;
            y[((size_t)samp[ls] * F + f) * D + c] = L.r[(size_t)t * L.ldd + c];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// Backward kernel
// ------------------------------------------------------------------------------------------------------
// slab layout (floats) of one workgroup:
//   [wq D*D][wk D*D][wv D*D][wo D*D][w1q D*U][w2q U*D][w1k D*U][w2k U*D][ln 2D][lnq 2D][lnk 2D]
struct SlabOff {
    int wq, wk, wv, wo, w1q, w2q, w1k, w2k, ln, lnq, lnk, total;
};
__host__ __device__ inline SlabOff slab_offsets(int D, int U) {
    SlabOff s;
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

// slab[i*N + o] (+)= sum_t A[t][i] * G[t][o]   for i < M, o < N
// MFMA form: the contraction runs over TOKENS (4 per MFMA step), both operands are 4-byte reads of the
// [token][feature] LDS images (A[i = lane&15][k = token 4*kt + (lane>>4)], B[k][j = lane&15]); one work item is a
// 16 x 16 block of the weight gradient.  Rows >= ntok are masked to zero.
__device__ __forceinline__ void outer_accumulate(bool mfma, float* __restrict__ slab, bool first, const float* A,
                                                 int lda, const float* G, int ldg, int M, int N, int ntok) {
    if (mfma) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
        const int n = lane & 15, g = lane >> 4;
        const int nmb = M >> 4, nnb = N >> 4;
        for (int item = wave; item < nmb * nnb; item += nw) {
            const int mt = item / nnb, nt = item - mt * nnb;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int t0 = 0; t0 < ntok; t0 += 4) {
                const int tok = t0 + g;
                const bool ok = tok < ntok;
                const float av = ok ? A[(size_t)tok * lda + 16 * mt + n] : 0.f;
                const float gv = ok ? G[(size_t)tok * ldg + 16 * nt + n] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, gv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = (16 * mt + 4 * g + r) * N + 16 * nt + n;
                slab[e] = first ? acc[r] : slab[e] + acc[r];
            }
        }
        return;
    }
    for (int e = threadIdx.x; e < M * N; e += blockDim.x) {
        const int i = e / N, o = e - i * N;
        float acc = 0.f;
        for (int t = 0; t < ntok; ++t) acc = fmaf(A[(size_t)t * lda + i], G[(size_t)t * ldg + o], acc);
        slab[e] = first ? acc : slab[e] + acc;
    }
}

// LayerNorm backward of a tile.  On entry z holds the pre-norm rows and g the gradient of the normalised
// output; on exit z holds the normalised rows z_hat, g holds the gradient wrt the pre-norm rows, and the
// slab entries [gamma D | beta D] have received this tile's contribution.
__device__ void layer_norm_backward_tile(float* z, float* g, int ld, int ntok, int D, const float* __restrict__ gamma,
                                         float* rstd_buf, float* __restrict__ slab, bool first) {
    for (int t = threadIdx.x; t < ntok; t += blockDim.x) {
        float mean, rstd;
        row_stats(z + (size_t)t * ld, D, mean, rstd);
        for (int c = 0; c < D; ++c) z[(size_t)t * ld + c] = (z[(size_t)t * ld + c] - mean) * rstd;
        rstd_buf[t] = rstd;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * D; e += blockDim.x) {
        const int c = e % D;
        float acc = 0.f;
        if (e < D) {
            for (int t = 0; t < ntok; ++t) acc = fmaf(g[(size_t)t * ld + c], z[(size_t)t * ld + c], acc);
        } else {
            for (int t = 0; t < ntok; ++t) acc += g[(size_t)t * ld + c];
        }
        slab[e] = first ? acc : slab[e] + acc;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ntok; t += blockDim.x) {
        float* gr = g + (size_t)t * ld;
        const float* zr = z + (size_t)t * ld;
        float m1 = 0.f, m2 = 0.f;
        for (int c = 0; c < D; ++c) {
            const float gg = gr[c] * gamma[c];
            m1 += gg;
            m2 = fmaf(gg, zr[c], m2);
        }
        m1 /= (float)D;
        m2 /= (float)D;
        const float rstd = rstd_buf[t];
        for (int c = 0; c < D; ++c) gr[c] = rstd * (gr[c] * gamma[c] - m1 - zr[c] * m2);
    }
    __syncthreads();
}

// Backward of one MetaNet role.  On entry g = gradient wrt the MetaNet output (post-LN), z = pre-LN rows,
// h = post-ReLU hidden, in0 = MetaNet input (q0/k0).  On exit g = gradient wrt the MetaNet input.
__device__ void metanet_backward_tile(const satrans_layer_desc& a, const TileDims& T, const int32_t* samp,
                                      const DropCtx& dc, int site, const float* __restrict__ tab_row,
                                      const float* gamma, const float* in0, float* h, float* z, float* g, float* gm,
                                      float* rstd_buf, LayerLds& L, float* slab_w1, float* slab_w2, float* slab_ln,
                                      bool first) {
    const int D = T.D, U = T.U, F = T.F, ntok = T.ntok;
    layer_norm_backward_tile(z, g, L.ldd, ntok, D, gamma, rstd_buf, slab_ln, first);
    // gm = dropout mask * dz
    for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
        const int t = i / D, c = i - t * D;
        const int ls = t / F, f = t - ls * F;
        gm[(size_t)t * L.ldd + c] = g[(size_t)t * L.ldd + c] * drop_mask(dc, site, samp[ls], (uint32_t)(f * D + c));
    }
    float* w2t = L.w;                        // [D][U + wp]: w2t[o][u] = W2[u][o]
    float* w1t = L.w + D * (U + L.wp);       // [U][D + wp]: w1t[u][i] = W1[i][u]
    stage_transposed(tab_row + D * U, w2t, U, D, U + L.wp);
    __syncthreads();
    outer_accumulate(L.mfma, slab_w2, first, h, L.ldu, gm, L.ldd, U, D, ntok);  // dW2[u][o] += h^T gm
    __syncthreads();
    // dh = (gm @ W2^T) * [h > 0], in place of h
    gemm_tokens<2>(L.mfma, gm, L.ldd, w2t, U + L.wp, D, U, ntok, [&](int t, int u, float acc) {
        float& hv = h[(size_t)t * L.ldu + u];
        hv = hv > 0.f ? acc : 0.f;
    });
    stage_transposed(tab_row, w1t, D, U, D + L.wp);
    __syncthreads();
    outer_accumulate(L.mfma, slab_w1, first, in0, L.ldd, h, L.ldu, D, U, ntok);  // dW1[i][u] += in0^T dh
    // g = dz + dh @ W1^T
    gemm_tokens<2>(L.mfma, h, L.ldu, w1t, D + L.wp, U, D, ntok,
                   [&](int t, int i, float acc) { g[(size_t)t * L.ldd + i] += acc; });
    __syncthreads();
}

// LDS aliasing (each pair is separated by a barrier between the last read of the first and the first write of the
// second): du / MetaNet dm scratch lives in r (dead once the final LayerNorm backward has consumed it), the
// gradient of the attention output replaces the attention output o (dead after dWo), dv replaces v (last read by
// the softmax-backward pass).
__global__ __launch_bounds__(kLayerBlock) void layer_bwd_kernel(satrans_layer_desc a, int Tsamp, int use_mfma,
                                                              const float* __restrict__ dy, float* __restrict__ dx,
                                                              float* __restrict__ slabs) {
    extern __shared__ __align__(16) float lds[];
    const int F = a.F, D = a.D, H = a.H, U = a.U, d = D / H;
    const int scen = blockIdx.y;
    TileDims T{F, D, H, U, d, 0, 0};
    LayerLds L;
    L.mfma = use_mfma;
    L.wp = use_mfma ? 4 : 0;
    L.ldd = D + (use_mfma ? 4 : 1);
    L.ldu = U + (use_mfma ? 4 : 1);
    L.ldp = F + 1;
    const int maxtok = Tsamp * F;
    const int nd = maxtok * L.ldd, nu = max(maxtok * L.ldu, maxtok * L.ldd), np = Tsamp * H * F * L.ldp;
    float* p = lds;
    auto take = [&](int n) { float* r = p; p += (n + 3) & ~3; return r; };
    L.x = take(nd); L.q0 = take(nd); L.k0 = take(nd); L.v = take(nd);
    L.zq = take(nd); L.q = take(nd); L.zk = take(nd); L.k = take(nd);
    L.o = take(nd); L.r = take(nd);
    L.u = (a.flags & SATRANS_RELU_OUT) ? take(nd) : nullptr;
    L.hq = take(nu); L.hk = take(nu);
    L.P = take(np);
    float* dS = take(np);
    float* g_r = take(nd);   // dy -> dr -> dx
    float* g_q = take(nd);
    float* g_k = take(nd);
    float* rstd_buf = take(maxtok);
    L.w = take(max(3 * D * (D + L.wp), D * (U + L.wp) + U * (D + L.wp)));
    float* g_m = L.r;        // du / MetaNet dm scratch
    float* g_o = L.o;        // gradient of the attention output
    float* g_v = L.v;
    const bool q_mod = (a.flags & SATRANS_BILINEAR) || (a.flags & SATRANS_META_Q);
    const bool k_mod = !(a.flags & SATRANS_BILINEAR) && (a.flags & SATRANS_META_K);
    if (!q_mod) { L.zq = L.q0; L.q = L.q0; }
    if (!k_mod) { L.zk = L.k0; L.k = L.k0; }

    const SlabOff so = slab_offsets(D, U);
    float* slab = slabs + ((size_t)scen * gridDim.x + blockIdx.x) * so.total;
    const DropCtx dc = make_drop(a);
    const float sqrt_d = sqrtf((float)d);
    bool first = true;

    for (int tile = blockIdx.x;; tile += gridDim.x) {
        const int32_t* samp;
        if (!tile_of(a, scen, tile, Tsamp, samp, T.nS)) break;
        T.ntok = T.nS * F;
        const int ntok = T.ntok;
        forward_tile(a, T, scen, samp, dc, L);
        for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
            const int t = i / D, c = i - t * D;
            const int ls = t / F, f = t - ls * F;
            g_r[(size_t)t * L.ldd + c] = dy[((size_t)samp[ls] * F + f) * D + c];
        }
        __syncthreads();
        // ---- final LayerNorm backward: g_r becomes dr ---------------------------------------------------
        layer_norm_backward_tile(L.r, g_r, L.ldd, ntok, D, a.ln_g, rstd_buf, slab + so.ln, first);
        // ---- du = dr * dropout mask * relu mask ------------------------------------------------------------
        for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
            const int t = i / D, c = i - t * D;
            const int ls = t / F, f = t - ls * F;
            float g = g_r[(size_t)t * L.ldd + c] * drop_mask(dc, kSiteOut, samp[ls], (uint32_t)(f * D + c));
            if ((a.flags & SATRANS_RELU_OUT) && !(L.u[(size_t)t * L.ldd + c] > 0.f)) g = 0.f;
            g_m[(size_t)t * L.ldd + c] = g;
        }
        stage_rows(a.w_out, L.w, D, D, D + L.wp);  // as [K = out feature][N = in feature]
        __syncthreads();
        outer_accumulate(L.mfma, slab + so.wo, first, g_m, L.ldd, L.o, L.ldd, D, D, ntok);  // dWo[o][i] += du^T o_att
        __syncthreads();                                                                // o is dead: g_o takes its place
        gemm_tokens<2>(L.mfma, g_m, L.ldd, L.w, D + L.wp, D, D, ntok,
                       [&](int t, int i, float acc) { g_o[(size_t)t * L.ldd + i] = acc; });
        __syncthreads();
        // ---- attention backward: dS rows (softmax + dropout backward) ---------------------------------------
        for (int task = threadIdx.x; task < T.nS * H * F; task += blockDim.x) {
            const int ls = task / (H * F), rem = task - ls * H * F;
            const int h = rem / F, qi = rem - h * F;
            const int b = samp[ls];
            const float* go = g_o + (size_t)(ls * F + qi) * L.ldd + h * d;
            const float* prow = L.P + (size_t)task * L.ldp;
            float* srow = dS + (size_t)task * L.ldp;
            float dot = 0.f;
            for (int j = 0; j < F; ++j) {
                const float* vj = L.v + (size_t)(ls * F + j) * L.ldd + h * d;
                float dp = 0.f;
                for (int e = 0; e < d; ++e) dp = fmaf(go[e], vj[e], dp);
                dp *= drop_mask(dc, kSiteAttn, b, drop_attn_elem(h, F, qi, j));
                srow[j] = dp;
                dot = fmaf(dp, prow[j], dot);
            }
            for (int j = 0; j < F; ++j) srow[j] = prow[j] * (srow[j] - dot) / sqrt_d;
        }
        __syncthreads();
        // ---- dq, dk, dv -----------------------------------------------------------------------------------------
        for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
            const int t = i / D, c = i - t * D;
            const int ls = t / F, row = t - ls * F, h = c / d;
            const int b = samp[ls];
            const int task0 = (ls * H + h) * F;
            float aq = 0.f, ak = 0.f, av = 0.f;
            for (int j = 0; j < F; ++j) {
                // dq[row] = sum_j dS[row][j] k[j] ; dk[row] = sum_i dS[i][row] q[i] ; dv[row] = sum_i Pd[i][row] do[i]
                aq = fmaf(dS[(size_t)(task0 + row) * L.ldp + j], L.k[(size_t)(ls * F + j) * L.ldd + c], aq);
                ak = fmaf(dS[(size_t)(task0 + j) * L.ldp + row], L.q[(size_t)(ls * F + j) * L.ldd + c], ak);
                const float pd = L.P[(size_t)(task0 + j) * L.ldp + row] *
                                 drop_mask(dc, kSiteAttn, b, drop_attn_elem(h, F, j, row));
                av = fmaf(pd, g_o[(size_t)(ls * F + j) * L.ldd + c], av);
            }
            g_q[(size_t)t * L.ldd + c] = aq;
            g_k[(size_t)t * L.ldd + c] = ak;
            g_v[(size_t)t * L.ldd + c] = av;
        }
        __syncthreads();
        // ---- modulation backward (g_q, g_k become gradients wrt q0, k0) ---------------------------------------------------
        if (a.flags & SATRANS_GATE) {
            // q = q0 * vec * 2 : d vec[c] += sum_t g[t][c] * 2 q0[t][c] (first D entries of the role's slab region), g *= 2 vec
            const float* gq = a.tab_q + (size_t)scen * a.tab_stride;
            const float* gk = a.tab_k + (size_t)scen * a.tab_stride;
            for (int e = threadIdx.x; e < 2 * D; e += blockDim.x) {
                const bool role_k = e >= D;
                const int c = role_k ? e - D : e;
                if (!(a.flags & (role_k ? SATRANS_META_K : SATRANS_META_Q))) continue;
                const float* g = role_k ? g_k : g_q;
                const float* in0 = role_k ? L.k0 : L.q0;
                float acc = 0.f;
                for (int t = 0; t < ntok; ++t) acc = fmaf(g[(size_t)t * L.ldd + c], 2.0f * in0[(size_t)t * L.ldd + c], acc);
                float* dst = slab + (role_k ? so.w1k : so.w1q) + c;
                *dst = first ? acc : *dst + acc;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
                const int t = i / D, c = i - t * D;
                const size_t at = (size_t)t * L.ldd + c;
                if (a.flags & SATRANS_META_Q) g_q[at] *= 2.0f * gq[c];
                if (a.flags & SATRANS_META_K) g_k[at] *= 2.0f * gk[c];
            }
            __syncthreads();
        } else if (a.flags & SATRANS_BILINEAR) {
            // q_h = q0_h @ M[h] : dM[h][e][e2] += sum_t q0[t][hd+e] g[t][hd+e2] ; dq0[t][hd+e] = sum_e2 g[t][hd+e2] M[h][e][e2]
            const float* M = a.tab_q + (size_t)scen * a.tab_stride;
            for (int e_ = threadIdx.x; e_ < D * d; e_ += blockDim.x) {
                const int h = e_ / (d * d), rem = e_ - h * d * d;
                const int e = rem / d, e2 = rem - e * d;
                float acc = 0.f;
                for (int t = 0; t < ntok; ++t)
                    acc = fmaf(L.q0[(size_t)t * L.ldd + h * d + e], g_q[(size_t)t * L.ldd + h * d + e2], acc);
                float* dst = slab + so.w1q + e_;
                *dst = first ? acc : *dst + acc;
            }
            for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
                const int t = i / D, c = i - t * D;
                const int h = c / d, e = c - h * d;
                const float* grow = g_q + (size_t)t * L.ldd + h * d;
                float acc = 0.f;
                for (int e2 = 0; e2 < d; ++e2) acc = fmaf(grow[e2], M[(h * d + e) * d + e2], acc);
                g_m[(size_t)t * L.ldd + c] = acc;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
                const int t = i / D, c = i - t * D;
                g_q[(size_t)t * L.ldd + c] = g_m[(size_t)t * L.ldd + c];
            }
            __syncthreads();
        } else {
            if (a.flags & SATRANS_META_Q)
                metanet_backward_tile(a, T, samp, dc, kSiteMetaQ, a.tab_q + (size_t)scen * a.tab_stride, a.lnq_g, L.q0,
                                      L.hq, L.zq, g_q, g_m, rstd_buf, L, slab + so.w1q, slab + so.w2q, slab + so.lnq, first);
            if (a.flags & SATRANS_META_K)
                metanet_backward_tile(a, T, samp, dc, kSiteMetaK, a.tab_k + (size_t)scen * a.tab_stride, a.lnk_g, L.k0,
                                      L.hk, L.zk, g_k, g_m, rstd_buf, L, slab + so.w1k, slab + so.w2k, slab + so.lnk, first);
        }
        // ---- projections: dW{q,k,v} += x^T g ; dx = dr*res + g_q Wq^T + g_k Wk^T + g_v Wv^T ----------------------
        outer_accumulate(L.mfma, slab + so.wq, first, L.x, L.ldd, g_q, L.ldd, D, D, ntok);
        outer_accumulate(L.mfma, slab + so.wk, first, L.x, L.ldd, g_k, L.ldd, D, D, ntok);
        outer_accumulate(L.mfma, slab + so.wv, first, L.x, L.ldd, g_v, L.ldd, D, D, ntok);
        const int ldt = D + L.wp, wsz = D * ldt;
        stage_transposed(a.w_query, L.w, D, D, ldt);              // w[o][i] = Wq[i][o]
        stage_transposed(a.w_key, L.w + wsz, D, D, ldt);
        stage_transposed(a.w_value, L.w + 2 * wsz, D, D, ldt);
        if (a.flags & SATRANS_NO_RES) {
            for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) g_r[(size_t)(i / D) * L.ldd + (i % D)] = 0.f;
        }
        __syncthreads();
        gemm_tokens<2>(L.mfma, g_q, L.ldd, L.w, ldt, D, D, ntok,
                       [&](int t, int i, float acc) { g_r[(size_t)t * L.ldd + i] += acc; });
        gemm_tokens<2>(L.mfma, g_k, L.ldd, L.w + wsz, ldt, D, D, ntok,
                       [&](int t, int i, float acc) { g_r[(size_t)t * L.ldd + i] += acc; });
        gemm_tokens<2>(L.mfma, g_v, L.ldd, L.w + 2 * wsz, ldt, D, D, ntok,
                       [&](int t, int i, float acc) { g_r[(size_t)t * L.ldd + i] += acc; });
        __syncthreads();
        for (int i = threadIdx.x; i < ntok * D; i += blockDim.x) {
            const int t = i / D, c = i - t * D;
            const int ls = t / F, f = t - ls * F;
            dx[((size_t)samp[ls] * F + f) * D + c] = g_r[(size_t)t * L.ldd + c];
        }
        __syncthreads();
        first = false;
    }
    if (first) {  // no tile for this workgroup: its slab must still read as zeros
        for (int e = threadIdx.x; e < so.total; e += blockDim.x) slab[e] = 0.f;
    }
}

// First level of the slab reduction: partial[(s*G + grp)][e] = sum of the slabs of scenario s whose workgroup index
// falls in group grp (contiguous ranges, summed in index order).
__global__ void slab_group_sum_kernel(const float* __restrict__ slabs, int gx, int G, int total,
                                      float* __restrict__ partial) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int s = blockIdx.y / G, grp = blockIdx.y - s * G;
    const int per = (gx + G - 1) / G;
    const int lo = grp * per, hi = min(gx, lo + per);
    float acc = 0.f;
    for (int w = lo; w < hi; ++w) acc += slabs[((size_t)s * gx + w) * total + e];
    partial[(size_t)blockIdx.y * total + e] = acc;
}

// Sum the per-workgroup slabs in (scenario, workgroup) order and ADD them to the parameter gradients.
__global__ void layer_bwd_reduce_kernel(const float* __restrict__ slabs, int S, int gx, int D, int U, int H, int flags,
                                        int64_t tab_stride, float* g_wq, float* g_wk, float* g_wv, float* g_wo,
                                        float* g_ln, float* g_lnq, float* g_lnk, float* g_tab_q, float* g_tab_k) {
    const SlabOff so = slab_offsets(D, U);
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= so.total) return;
    const bool is_tab = e >= so.w1q && e < so.ln;
    if (!is_tab) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s)
            for (int w = 0; w < gx; ++w) acc += slabs[((size_t)s * gx + w) * so.total + e];
        if (e < so.wk) g_wq[e - so.wq] += acc;
        else if (e < so.wv) g_wk[e - so.wk] += acc;
        else if (e < so.wo) g_wv[e - so.wv] += acc;
        else if (e < so.w1q) g_wo[e - so.wo] += acc;
        else if (e < so.lnq) g_ln[e - so.ln] += acc;
        else if (flags & (SATRANS_GATE | SATRANS_BILINEAR)) return;   // no MetaNet, no MetaNet LayerNorm
        else if (e < so.lnk) {
            // Without 'pos' the Q and K MetaNets share ONE LayerNorm (satrans.py:46): g_lnq == g_lnk, and this
            // thread adds both roles' partials (Q first) so that no two threads update one address.
            const bool shared = (flags & SATRANS_META_Q) && (flags & SATRANS_META_K) && g_lnq == g_lnk;
            if (shared) {
                float ak = 0.f;
                for (int s = 0; s < S; ++s)
                    for (int w = 0; w < gx; ++w) ak += slabs[((size_t)s * gx + w) * so.total + e + 2 * D];
                g_lnq[e - so.lnq] += acc + ak;
            } else if (flags & SATRANS_META_Q) {
                g_lnq[e - so.lnq] += acc;
            }
        } else {
            const bool shared = (flags & SATRANS_META_Q) && (flags & SATRANS_META_K) && g_lnq == g_lnk;
            if ((flags & SATRANS_META_K) && !shared) g_lnk[e - so.lnk] += acc;
        }
        return;
    }
    // generated-weight row layout: [W1 D*U | W2 U*D]  (submodules.py:82-86).  The thread of a Q-role element also
    // folds in the K-role partial of the same generated weight, so that an aliased table (no 'pos' flag:
    // tab_q == tab_k) is updated by exactly one thread, Q contribution first.
    if (e >= so.w1k) return;
    const int within = e - so.w1q;
    const int ek = so.w1k + within;
    // length of the generated row that is actually used, and which roles read it
    bool mq = flags & SATRANS_META_Q, mk = flags & SATRANS_META_K;
    int tab_len = 2 * D * U;
    if (flags & SATRANS_GATE) tab_len = D;                               // satrans.py:162-163
    if (flags & SATRANS_BILINEAR) { tab_len = D * (D / H); mq = true; mk = false; }   // satrans.py:159-161, applied to Q only
    if (within >= tab_len) return;
    for (int s = 0; s < S; ++s) {
        float aq = 0.f, ak = 0.f;
        for (int w = 0; w < gx; ++w) {
            if (mq) aq += slabs[((size_t)s * gx + w) * so.total + e];
            if (mk) ak += slabs[((size_t)s * gx + w) * so.total + ek];
        }
        if (mq && mk && g_tab_q == g_tab_k) {
            g_tab_q[(size_t)s * tab_stride + within] += aq + ak;
        } else {
            if (mq) g_tab_q[(size_t)s * tab_stride + within] += aq;
            if (mk) g_tab_k[(size_t)s * tab_stride + within] += ak;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
constexpr int kLdsBudgetFwd = 78 * 1024;   // two workgroups per CU
constexpr int kLdsBudgetBwd = 156 * 1024;  // one workgroup per CU
constexpr int kReduceSplit = 16;           // first-level groups of the slab reduction

static int64_t r4(int64_t n) { return (n + 3) & ~(int64_t)3; }

static int64_t fwd_lds_floats(int T, int F, int D, int H, int U, int mfma) {
    const int64_t tok = (int64_t)T * F;
    const int pad = mfma ? 4 : 1, wp = mfma ? 4 : 0;
    const int64_t hp = std::max<int64_t>(std::max<int64_t>(tok * (U + pad), tok * (D + pad)), (int64_t)T * H * F * (F + 1));
    const int64_t w = std::max<int64_t>((int64_t)D * (3 * D + wp), std::max<int64_t>((int64_t)D * (U + wp), (int64_t)U * (D + wp)));
    return 4 * r4(tok * (D + pad)) + r4(hp) + w + 64;
}

static int64_t bwd_lds_floats(int T, int F, int D, int H, int U, int flags, int mfma) {
    const int64_t tok = (int64_t)T * F;
    const int pad = mfma ? 4 : 1, wp = mfma ? 4 : 0;
    const int64_t nd = r4(tok * (D + pad)), nu = r4(std::max<int64_t>(tok * (U + pad), tok * (D + pad))),
                  np = r4((int64_t)T * H * F * (F + 1));
    const int nbuf = 13 + ((flags & SATRANS_RELU_OUT) ? 1 : 0);
    const int64_t w = std::max<int64_t>(3 * (int64_t)D * (D + wp), (int64_t)D * (U + wp) + (int64_t)U * (D + wp));
    return nbuf * nd + 2 * nu + 2 * np + r4(tok) + w + 64;
}

// MFMA needs 16-aligned feature counts; SATRANS_LAYER_IMPL=lds forces the scalar arm (ablation / debugging)
static int want_mfma(const satrans_layer_desc* d) {
    if (satrans_layer_impl() == 1) return 0;
    if (d->D % 16) return 0;
    if (!(d->flags & (SATRANS_GATE | SATRANS_BILINEAR)) && (d->flags & (SATRANS_META_Q | SATRANS_META_K)) && (d->U % 16))
        return 0;
    return 1;
}

static int validate(const satrans_layer_desc* d, const char* who) {
    SATRANS_REQUIRE(d, SATRANS_E_BADARG, "%s: null descriptor", who);
    SATRANS_REQUIRE(d->x && d->sid && d->order && d->seg && d->w_query && d->w_key && d->w_value && d->w_out &&
                        d->ln_g && d->ln_b,
                    SATRANS_E_BADARG, "%s: null tensor pointer", who);
    SATRANS_REQUIRE(d->B > 0 && d->F > 0 && d->D > 0 && d->H > 0 && d->S > 0, SATRANS_E_BADARG, "%s: bad sizes", who);
    SATRANS_REQUIRE(d->D % 4 == 0, SATRANS_E_UNSUPPORTED, "%s: embedding_size %d is not a multiple of 4", who, d->D);
    SATRANS_REQUIRE(d->D % d->H == 0, SATRANS_E_BADARG, "%s: embedding_size %d is not a multiple of head_num %d", who,
                    d->D, d->H);
    const bool gate = d->flags & SATRANS_GATE, bil = d->flags & SATRANS_BILINEAR;
    SATRANS_REQUIRE(!(gate && bil), SATRANS_E_BADARG, "%s: 'gate' and 'bilinear' together", who);
    if (gate || bil) {
        SATRANS_REQUIRE(d->U > 0 && d->tab_q && d->tab_k, SATRANS_E_BADARG, "%s: generated-weight table missing", who);
        const int64_t need = gate ? d->D : (int64_t)d->D * (d->D / d->H);
        SATRANS_REQUIRE(d->tab_stride >= need, SATRANS_E_BADARG, "%s: tab_stride %lld < %lld", who,
                        (long long)d->tab_stride, (long long)need);
        SATRANS_REQUIRE(2 * (int64_t)d->D * d->U >= need, SATRANS_E_UNSUPPORTED, "%s: U too small for the gradient slab", who);
    } else if (d->flags & (SATRANS_META_Q | SATRANS_META_K)) {
        SATRANS_REQUIRE(d->U > 0 && d->tab_q && d->tab_k && d->lnq_g && d->lnq_b && d->lnk_g && d->lnk_b,
                        SATRANS_E_BADARG, "%s: MetaNet tensors missing", who);
        SATRANS_REQUIRE(d->tab_stride >= 2 * (int64_t)d->D * d->U, SATRANS_E_BADARG, "%s: tab_stride too small", who);
    }
    SATRANS_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, SATRANS_E_BADARG, "%s: drop_p %f", who, d->drop_p);
    return SATRANS_OK;
}

struct LayerPlan {
    int T;        // samples per tile
    int gx;       // workgroups per scenario
    int mfma;
    size_t lds;   // dynamic LDS bytes
};

static int plan_fwd(const satrans_layer_desc* d, LayerPlan& p) {
    p.mfma = want_mfma(d);
    int T = 0;
    for (int t = 1; t <= 16; ++t)
        if (fwd_lds_floats(t, d->F, d->D, d->H, d->U, p.mfma) * 4 <= kLdsBudgetFwd) T = t;
    if (T == 0 && fwd_lds_floats(1, d->F, d->D, d->H, d->U, p.mfma) * 4 <= kLdsBudgetBwd) T = 1;
    SATRANS_REQUIRE(T > 0, SATRANS_E_UNSUPPORTED, "layer_fwd: one sample (F=%d D=%d U=%d) does not fit LDS", d->F, d->D,
                    d->U);
    p.T = T;
    p.lds = (size_t)fwd_lds_floats(T, d->F, d->D, d->H, d->U, p.mfma) * 4;
    const int64_t tiles = ceil_div(d->B, T);
    p.gx = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, ceil_div(256 * 4, d->S)));
    return SATRANS_OK;
}

static int plan_bwd(const satrans_layer_desc* d, LayerPlan& p) {
    p.mfma = want_mfma(d);
    int T = 0;
    for (int t = 1; t <= 8; ++t)
        if (bwd_lds_floats(t, d->F, d->D, d->H, d->U, d->flags, p.mfma) * 4 <= kLdsBudgetBwd) T = t;
    SATRANS_REQUIRE(T > 0, SATRANS_E_UNSUPPORTED, "layer_bwd: one sample (F=%d D=%d U=%d) does not fit LDS", d->F, d->D,
                    d->U);
    if (!p.mfma) {  // scalar arm: prefer two workgroups per CU when a smaller tile allows it
        for (int t = T; t >= 1; --t)
            if (bwd_lds_floats(t, d->F, d->D, d->H, d->U, d->flags, 0) * 4 <= kLdsBudgetFwd) { T = t; break; }
    }
    p.T = T;
    p.lds = (size_t)bwd_lds_floats(T, d->F, d->D, d->H, d->U, d->flags, p.mfma) * 4;
    const int64_t tiles = ceil_div(d->B, T);
    const int per_cu = p.lds * 2 <= (size_t)160 * 1024 ? 2 : 1;
    p.gx = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, ceil_div(256 * per_cu, d->S)));
    return SATRANS_OK;
}

}  // namespace satrans

using namespace satrans;

// two-level, fixed-order reduction of S*gx per-workgroup slabs (kReduceSplit contiguous groups per scenario first);
// the first-level partials live behind the slabs in the same workspace
extern "C" int64_t satrans_layer_slab_reduce_extra_floats(int S, int D, int U) {
    return (int64_t)S * kReduceSplit * slab_offsets(D, U).total;
}

extern "C" int satrans_layer_slab_reduce(float* slabs, int S, int gx, int D, int U, int H, int flags, int64_t tab_stride,
                                         float* g_wq, float* g_wk, float* g_wv, float* g_wo, float* g_ln, float* g_lnq,
                                         float* g_lnk, float* g_tab_q, float* g_tab_k, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int total = slab_offsets(D, U).total;
    float* partial = slabs + (size_t)S * gx * total;
    slab_group_sum_kernel<<<dim3((unsigned)ceil_div(total, 256), S * kReduceSplit), 256, 0, stream>>>(
        slabs, gx, kReduceSplit, total, partial);
    SATRANS_CHECK_LAUNCH("slab_group_sum_kernel");
    layer_bwd_reduce_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, stream>>>(
        partial, S, kReduceSplit, D, U, H, flags, tab_stride, g_wq, g_wk, g_wv, g_wo, g_ln, g_lnq, g_lnk, g_tab_q, g_tab_k);
    SATRANS_CHECK_LAUNCH("layer_bwd_reduce_kernel");
    return SATRANS_OK;
}

extern "C" int satrans_layer_validate(const satrans_layer_desc* d, const char* who) { return validate(d, who); }

extern "C" int satrans_layer_fwd_lds(const satrans_layer_desc* d, float* y, float* att, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = validate(d, "layer_fwd");
    if (rc) return rc;
    SATRANS_REQUIRE(y, SATRANS_E_BADARG, "layer_fwd: null output");
    LayerPlan p;
    rc = plan_fwd(d, p);
    if (rc) return rc;
    static size_t attr_set = 0;
    if (p.lds > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)layer_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)p.lds);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "layer_fwd: LDS attribute: %s", hipGetErrorString(e));
        attr_set = p.lds;
    }
    layer_fwd_kernel<<<dim3(p.gx, d->S), kLayerBlock, p.lds, stream>>>(*d, p.T, p.mfma, y, att);
    SATRANS_CHECK_LAUNCH("layer_fwd_kernel");
    return SATRANS_OK;
}

extern "C" int64_t satrans_layer_bwd_slab_floats_lds(const satrans_layer_desc* d) {
    LayerPlan p;
    if (!d || validate(d, "layer_bwd") || plan_bwd(d, p)) return -1;
    return ((int64_t)d->S * p.gx + (int64_t)d->S * kReduceSplit) * slab_offsets(d->D, d->U).total;
}

extern "C" int satrans_layer_bwd_lds(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, float* g_wq,
                                     float* g_wk, float* g_wv, float* g_wo, float* g_ln, float* g_lnq, float* g_lnk,
                                     float* g_tab_q, float* g_tab_k, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = validate(d, "layer_bwd");
    if (rc) return rc;
    SATRANS_REQUIRE(dy && dx && slabs && g_wq && g_wk && g_wv && g_wo && g_ln, SATRANS_E_BADARG, "layer_bwd: null pointer");
    const bool plain_meta = !(d->flags & (SATRANS_GATE | SATRANS_BILINEAR));
    if (d->flags & SATRANS_META_Q) SATRANS_REQUIRE((g_lnq || !plain_meta) && g_tab_q, SATRANS_E_BADARG, "layer_bwd: null Q gradient");
    if (d->flags & SATRANS_META_K) SATRANS_REQUIRE((g_lnk || !plain_meta) && g_tab_k, SATRANS_E_BADARG, "layer_bwd: null K gradient");
    if (d->flags & SATRANS_BILINEAR) SATRANS_REQUIRE(g_tab_q, SATRANS_E_BADARG, "layer_bwd: null bilinear gradient");
    LayerPlan p;
    rc = plan_bwd(d, p);
    if (rc) return rc;
    static size_t attr_set = 0;
    if (p.lds > attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)layer_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)p.lds);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "layer_bwd: LDS attribute: %s", hipGetErrorString(e));
        attr_set = p.lds;
    }
    layer_bwd_kernel<<<dim3(p.gx, d->S), kLayerBlock, p.lds, stream>>>(*d, p.T, p.mfma, dy, dx, slabs);
    SATRANS_CHECK_LAUNCH("layer_bwd_kernel");
    return satrans_layer_slab_reduce(slabs, d->S, p.gx, d->D, d->U, d->H, d->flags, d->tab_stride, g_wq, g_wk, g_wv, g_wo,
                                     g_ln, g_lnq, g_lnk, g_tab_q, g_tab_k, stream_);
}
