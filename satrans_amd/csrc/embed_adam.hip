// Embedding-table gradient + optimizer path.
//
// Reference semantics: the tables are dense parameters (`sparse=False`,
// models/meta_basemodel.py:168) regularised by sum(l2 * w^2) over EVERY row (models/meta_basemodel.py:577-593) and
// stepped by a dense torch.optim.Adam (main.py:343).  So every row of every table moves on every step: a row that
// was not gathered still sees the gradient 2*l2*w.  The reference pays for that with a table-sized gradient buffer
// (zero fill + index_add), a table-sized regulariser pass and a dense Adam pass (~9 table sweeps).  Here:
//
//   * gathered ("touched") rows: their gradient rows are grouped by a stable radix sort of the arena row ids and
//     summed in position order (bitwise reproducible, identical on every data-parallel rank), then stepped;
//   * all other rows: ONE streaming kernel that reads p,m,v and writes p,m,v (6 table sweeps, the algorithmic
//     minimum for exact dense-Adam semantics) and computes the regulariser sum on the way.  It does not depend on
//     the backward pass at all, only on the touched-row bitmap, so it can overlap the attention kernels.
//
// Adam arithmetic follows torch.optim.adam._single_tensor_adam (lerp for exp_avg, mul+addcmul for exp_avg_sq,
// denom = sqrt(v)/sqrt(bc2) + eps, addcdiv with -lr/bc1).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace satrans {

constexpr int kChunk = 16;         // sorted positions per lane group in the touched-row pass (A/B on one box, round 4: 8 / 16 / 32 / 64 -> 75 / 74 / 81 / 96 us for the five launches)
constexpr int kStreamBlock = 256;
constexpr int kStreamBlocks = 2048;  // upper bound of the streaming grid = slots reserved for its partial sums

struct AdamK {
    float neg_step, w1, beta2, w2, eps, l2x2, l2;
    double rbc2;   // 1 / (double)bc2_sqrt, see adam_core
    float rbc2f;   // ... rounded to fp32: the fast arithmetic's factor
    int fast;      // satrans_adam_hparams.arith: 0 = torch's operations bit for bit, 1 = hardware sqrt / reciprocal (adam_core)
};

__host__ inline AdamK make_adamk(const satrans_adam_hparams& h) {
    AdamK k;
    k.neg_step = -h.lr_over_bc1;
    k.rbc2 = 1.0 / (double)h.bc2_sqrt;
    k.w1 = (float)(1.0 - (double)h.beta1);
    k.beta2 = h.beta2;
    k.w2 = (float)(1.0 - (double)h.beta2);
    k.eps = h.eps;
    k.l2 = h.l2;
    k.l2x2 = 2.0f * h.l2;
    k.rbc2f = (float)k.rbc2;
    k.fast = h.arith == SATRANS_ADAM_FAST ? 1 : 0;
    return k;
}

// One Adam update of one element.  Every product / sum is spelled out (fmaf, __fmul_rn, ...) so that hipcc cannot
// contract the expression differently in different kernels: the streaming kernel, the touched-row kernels and the
// lazy replay must produce bit-identical tables.
// sqrt(v) / bc2_sqrt is a division by a per-step constant c.  It is evaluated as RN32((double)sqrt(v) * RN64(1/c)), which is the
// correctly rounded fp32 quotient (so bit for bit torch's division): for fp32 s and c the quotient s/c lies at a relative
// distance > 2^-49 from every rounding boundary of fp32 (a boundary m has a 25-bit significand, and s - m*c is a non-zero
// multiple of the last place of m*c, whose significand has at most 49 bits), while the double product is within 2^-52 of s/c;
// results are never subnormal here (sqrt(v) >= 3.7e-23, c <= 1).  Four instructions instead of the eleven of an IEEE
// division - the lazy replay and flush kernels are bound by exactly this arithmetic.
__device__ __forceinline__ void adam_core(float& p, float& m, float& v, float g, float neg_step, double rbc2, float w1,
                                          float beta2, float w2, float eps) {
    m = fmaf(w1, __fsub_rn(g, m), m);                               // exp_avg.lerp_(grad, 1 - beta1), weight < 0.5 branch
    v = fmaf(__fmul_rn(w2, g), g, __fmul_rn(v, beta2));             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = __fadd_rn((float)__dmul_rn((double)sqrtf(v), rbc2), eps);   // sqrt(v) / bc2_sqrt + eps
    p = __fadd_rn(p, __fdiv_rn(__fmul_rn(neg_step, m), denom));     // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// The FAST arithmetic (satrans_adam_hparams.arith = SATRANS_ADAM_FAST, the engine's default since round 6): the moments as
// above, the update through the hardware's square root and reciprocal as they are (v_sqrt_f32, v_rcp_f32: 1 ulp each; a
// subnormal second moment counts as zero, < 1e-19 against eps) and one multiply by the fp32 reciprocal of bc2_sqrt:
//     denom = fma(sqrt_hw(v), rbc2f, eps);   p = fma(neg_step * m, rcp_hw(denom), p)
// Relative error of ONE update against torch's correctly rounded operations <= 2^-21.6 (3 roundings of half an ulp and two
// 1-ulp instructions), i.e. <= 3.2e-7 * |update| <= 3.2e-7 * lr per element and step - measured drift in INTEGRATION.md 4.
// The same instruction sequence in every kernel form (streaming, gathered rows, lazy replay / flush: packed fp32 operations
// are IEEE per element), so the forms still leave identical bits.  ~2.5x fewer VALU cycles per element-step than the
// correctly rounded sequences the exact form needs (its flush sits on their instruction floor: profiles/r05_valu_rates.txt).
__device__ __forceinline__ void adam_core_fast(float& p, float& m, float& v, float g, float neg_step, float rbc2f, float w1,
                                               float beta2, float w2, float eps) {
    m = fmaf(w1, __fsub_rn(g, m), m);
    v = fmaf(__fmul_rn(w2, g), g, __fmul_rn(v, beta2));
    const float denom = fmaf(__builtin_amdgcn_sqrtf(v), rbc2f, eps);
    p = fmaf(__fmul_rn(neg_step, m), __builtin_amdgcn_rcpf(denom), p);
}

__device__ __forceinline__ void adam1(float& p, float& m, float& v, float g, const AdamK& k) {
    if (k.fast) adam_core_fast(p, m, v, g, k.neg_step, k.rbc2f, k.w1, k.beta2, k.w2, k.eps);      // (uniform: a scalar branch)
    else adam_core(p, m, v, g, k.neg_step, k.rbc2, k.w1, k.beta2, k.w2, k.eps);
}

__device__ __forceinline__ double adam4(float4& p, float4& m, float4& v, float4 g, const AdamK& k) {
    const double reg = (double)k.l2 * ((double)p.x * p.x + (double)p.y * p.y + (double)p.z * p.z + (double)p.w * p.w);
    adam1(p.x, m.x, v.x, __fadd_rn(g.x, __fmul_rn(k.l2x2, p.x)), k);
    adam1(p.y, m.y, v.y, __fadd_rn(g.y, __fmul_rn(k.l2x2, p.y)), k);
    adam1(p.z, m.z, v.z, __fadd_rn(g.z, __fmul_rn(k.l2x2, p.z)), k);
    adam1(p.w, m.w, v.w, __fadd_rn(g.w, __fmul_rn(k.l2x2, p.w)), k);
    return reg;
}

// streaming (non-temporal) 16-byte accesses: the tables are far larger than L2 + Infinity Cache and every byte is
// used exactly once per step, so keeping them out of the caches leaves the cache to the attention kernels
using f4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ float4 nt_load(const float4* p) {
    const f4 t = __builtin_nontemporal_load((const f4*)p);
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nt_store(const float4& v, float4* p) {
    f4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, (f4*)p);
}

// block-wide sum of one double per thread, fixed tree order; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    scratch[threadIdx.x] = v;
    __syncthreads();
    for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) scratch[threadIdx.x] += scratch[threadIdx.x + off];
        __syncthreads();
    }
    return scratch[0];
}

// ---------------------------------------------------------------------------------------------------------
// untouched rows: the dominant kernel of a training step (HBM-bound, 6 table sweeps)
// ---------------------------------------------------------------------------------------------------------
template <int LPR, int UNR, bool NT, bool CHUNKED = false>
__global__ __launch_bounds__(kStreamBlock) void adam_untouched_kernel(float4* __restrict__ P, float4* __restrict__ M,
                                                                    float4* __restrict__ V, int64_t first4, int64_t n4,
                                                                    const uint32_t* __restrict__ touched, AdamK k,
                                                                    double* __restrict__ reg_partials) {
    __shared__ double s_red[kStreamBlock];
    double reg = 0.0;
    // elements [first4, n4): the rows before first4 / LPR belong to the small tables, which take a dense step of their own
    P += first4; M += first4; V += first4;
    n4 -= first4;
    const int64_t first_row = first4 / LPR;
    // CHUNKED: every block sweeps one contiguous slice of the arrays (DRAM-page friendly) instead of a grid-wide stride
    const int64_t stride = CHUNKED ? (int64_t)kStreamBlock : (int64_t)gridDim.x * kStreamBlock;
    const int64_t slice = (((n4 + gridDim.x - 1) / gridDim.x) + kStreamBlock * UNR - 1) / (kStreamBlock * UNR) * (kStreamBlock * UNR);
    const int64_t begin = CHUNKED ? (int64_t)blockIdx.x * slice + threadIdx.x : (int64_t)blockIdx.x * kStreamBlock + threadIdx.x;
    const int64_t limit = CHUNKED ? min(n4, ((int64_t)blockIdx.x + 1) * slice) : n4;
    for (int64_t i0 = begin; i0 < limit; i0 += stride * UNR) {
        float4 p[UNR], m[UNR], v[UNR];
        bool live[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = i0 + u * stride;
            live[u] = false;
            if (i < limit) {
                const int64_t row = first_row + i / LPR;
                live[u] = !((touched[row >> 5] >> (row & 31)) & 1u);
            }
            if (live[u]) {
                p[u] = NT ? nt_load(&P[i]) : P[i];
                m[u] = NT ? nt_load(&M[i]) : M[i];
                v[u] = NT ? nt_load(&V[i]) : V[i];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (live[u]) {
                const int64_t i = i0 + u * stride;
                reg += adam4(p[u], m[u], v[u], make_float4(0.f, 0.f, 0.f, 0.f), k);
                if (NT) { nt_store(p[u], &P[i]); nt_store(m[u], &M[i]); nt_store(v[u], &V[i]); }
                else { P[i] = p[u]; M[i] = m[u]; V[i] = v[u]; }
            }
        }
    }
    const double total = block_sum(reg, s_red);
    if (threadIdx.x == 0) reg_partials[blockIdx.x] = total;
}

// ---------------------------------------------------------------------------------------------------------
// touched rows: sort, segmented sum in position order, step
// ---------------------------------------------------------------------------------------------------------
__global__ void iota_kernel(const int32_t* __restrict__ rows, int64_t n, uint32_t* __restrict__ keys,
                            int32_t* __restrict__ pos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = (uint32_t)rows[i];
    pos[i] = (int32_t)i;
}

// Touched-row bitmap from the SORTED ids: only the first position of every run sets its bit, so a row gathered
// thousands of times (small-vocabulary fields) costs one atomic instead of thousands on one address.
__global__ void mark_heads_kernel(const int32_t* __restrict__ sorted_rows, int64_t n, uint32_t* __restrict__ touched) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = (uint32_t)sorted_rows[i];
    if (i > 0 && (uint32_t)sorted_rows[i - 1] == r) return;
    atomicOr(&touched[r >> 5], 1u << (r & 31));
}

// Second level: a run that crosses chunk boundaries is the ordered sum  trail(c0) + lead(c0+1) + ... + lead(c1).
// One lane group per SUPERCHUNK of kSuper chunks walks its chunks once: runs that end inside the superchunk take their
// optimizer step in the finish body; what crosses a superchunk boundary leaves a record of the same shape one level up (lead/trail
// partial sums, flags as for chunks), which the spans body finishes.  A row gathered by every sample of every
// rank (small-vocabulary fields: runs of tens of thousands of positions) thus costs kSuper + run/(kChunk*kSuper)
// dependent steps instead of run/kChunk.
// The walk runs at the tail of the chunk kernel wherever a workgroup's 256 / LPR chunks are whole superchunks (D <= 64: the
// records it reads were written by the same workgroup, one barrier earlier) - one dependent launch less in the chain - and as
// its own launch otherwise (D = 128).  (No __restrict__ here: in the fused form these are the arrays the chunk phase wrote.)
constexpr int kSuper = 16;
template <int LPR>
__device__ __forceinline__ void touched_super_body(int64_t group, int q, int64_t chunks, const float4* partial, const int32_t* info,
                                                   const int32_t* trail_row, float4* partial2, int32_t* info2, int32_t* trail_row2,
                                                   float4* done_sum, int32_t* done_row) {
    const int64_t c0 = group * kSuper;
    if (c0 < chunks) {
        const int64_t c1 = min(chunks, c0 + (int64_t)kSuper);
        // flags and lead partials of all chunks first (independent loads, issued together; a lead slot that was not written
        // this step holds stale numbers, which are read but never used)
        int f[kSuper];
        float4 lead[kSuper];
#pragma unroll
        for (int u = 0; u < kSuper; ++u) f[u] = c0 + u < c1 ? info[c0 + u] : 0;
#pragma unroll
        for (int u = 0; u < kSuper; ++u)
            lead[u] = c0 + u < c1 ? partial[((c0 + u) * 2 + 0) * LPR + q] : make_float4(0.f, 0.f, 0.f, 0.f);
        bool open = f[0] & 2, is_lead = open;
        int flags2 = open ? 2 : 0;
        int32_t row = -1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < kSuper; ++u) {
            const int64_t c = c0 + u;
            int32_t ended = -1;
            if (f[u] & 2) {
                const float4 g = lead[u];
                acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
                if (f[u] & 4) {
                    if (is_lead) {
                        partial2[(group * 2 + 0) * LPR + q] = acc;
                        flags2 |= 4;
                    } else {            // a run that ended in chunk c: its step is taken by the parallel finish body
                        done_sum[c * LPR + q] = acc;
                        ended = row;
                    }
                    open = false;
                    is_lead = false;
                }
            }
            if (f[u] & 1) {
                acc = partial[(c * 2 + 1) * LPR + q];
                row = trail_row[c];
                open = true;
                is_lead = false;
            }
            if (q == 0 && c < c1) done_row[c] = ended;
        }
        if (open) {
            if (is_lead) {
                partial2[(group * 2 + 0) * LPR + q] = acc;
            } else {
                partial2[(group * 2 + 1) * LPR + q] = acc;
                flags2 |= 1;
                if (q == 0) trail_row2[group] = row;
            }
        }
        if (q == 0) info2[group] = flags2;
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void touched_super_kernel(int64_t chunks, const float4* partial, const int32_t* info,
                                                           const int32_t* trail_row, float4* partial2, int32_t* info2,
                                                           int32_t* trail_row2, float4* done_sum, int32_t* done_row) {
    __builtin_amdgcn_s_setprio(3);      // (ahead of the side stream's next-batch kernels when they share a SIMD: 71 -> 62 us for the touched-row chain)
    touched_super_body<LPR>(((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR, threadIdx.x % LPR, chunks, partial, info, trail_row, partial2,
                            info2, trail_row2, done_sum, done_row);
}

// Per-chunk bookkeeping for segments that cross chunk boundaries.
//   info bit0: the chunk's LAST piece starts here and continues into the next chunk (a "trail" piece)
//   info bit1: the chunk's FIRST piece continues a segment from the previous chunk (a "lead" piece)
//   info bit2: that lead piece also ends inside this chunk (or at its end)
// lead/trail partial sums live in partial_ws: [chunks][2][D]
template <int LPR>
__global__ __launch_bounds__(256) void touched_chunks_kernel(float4* __restrict__ P, float4* __restrict__ M,
                                                            float4* __restrict__ V, const int32_t* __restrict__ sorted_rows,
                                                            const int32_t* __restrict__ src, int64_t n,
                                                            const float4* __restrict__ gemb, float4* partial,
                                                            int32_t* info, int32_t* trail_row,
                                                            float4* __restrict__ rowsum, int32_t* __restrict__ head_of,
                                                            float4* partial2, int32_t* info2, int32_t* trail_row2,
                                                            float4* done_sum, int32_t* done_row) {
    __builtin_amdgcn_s_setprio(3);      // (ahead of the side stream's next-batch kernels when they share a SIMD: 71 -> 62 us for the touched-row chain)
    const int64_t group = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
    const int q = threadIdx.x % LPR;
    const int64_t start = group * kChunk;
    if (start < n) {
        int64_t head = start;          // position where the current run started
        const int64_t end = min(n, start + (int64_t)kChunk);
        const int32_t prev_row = start > 0 ? sorted_rows[start - 1] : -1;
        const int32_t next_row = end < n ? sorted_rows[end] : -1;
        int32_t cur = sorted_rows[start];
        bool begins = cur != prev_row;
        int flags = 0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // a run that lies inside this chunk is complete: its sum goes to rowsum[head] and head_of[head] = 1; the
        // optimizer step itself is a separate, fully parallel launch (one dependent load->store chain per row here
        // would serialise ~25 HBM round trips per chunk)
        auto flush = [&](bool ends) {
            const bool complete = begins && ends;
            if (q == 0) head_of[head] = complete ? 1 : 0;
            if (complete) {
                rowsum[head * LPR + q] = acc;
            } else if (!begins) {
                partial[(group * 2 + 0) * LPR + q] = acc;
                flags |= 2 | (ends ? 4 : 0);
            } else {
                partial[(group * 2 + 1) * LPR + q] = acc;
                flags |= 1;
                if (q == 0) trail_row[group] = cur;
            }
        };
        // positions in batches of 8: the ids, source positions and the 8 gradient rows of a batch are loaded together
        // (8 independent 16-byte loads in flight per lane), then consumed in position order
        constexpr int kBatch = 16;      // (= kChunk: the whole chunk in one round of loads; 4 / 8 / 16 -> 74 / 73 / 69 us for the chain)
        for (int64_t jb = start; jb < end; jb += kBatch) {
            int32_t r[kBatch];
            float4 g[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int64_t j = jb + u;
                const bool ok = j < end;
                r[u] = ok ? sorted_rows[j] : -2;
                const int32_t sidx = ok ? src[j] : 0;
                g[u] = ok ? gemb[(int64_t)sidx * LPR + q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                if (jb + u < end) {
                    if (r[u] != cur) {
                        flush(true);
                        cur = r[u];
                        begins = true;
                        head = jb + u;
                        acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    if (q == 0 && jb + u != head) head_of[jb + u] = 0;
                    acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w;
                }
            }
        }
        flush(next_row != cur);
        if (q == 0) info[group] = flags;
    }
    // the superchunk walk over this workgroup's own chunk records (see touched_super_body)
    constexpr int kGroups = 256 / LPR;
    if constexpr (kGroups % kSuper == 0) {
        __syncthreads();
        const int sg = threadIdx.x / LPR;
        if (sg < kGroups / kSuper)
            touched_super_body<LPR>((int64_t)blockIdx.x * (kGroups / kSuper) + sg, q, (n + kChunk - 1) / kChunk, partial, info, trail_row,
                                    partial2, info2, trail_row2, done_sum, done_row);
    }
}

// Adam on every row whose run was complete inside one chunk: one lane group per sorted position.
// All three finishing kernels (this one, the superchunk and the spans kernel) have a second mode: with G != nullptr the
// finished sum of a row is STORED into the dense gradient buffer G[row] instead of being applied (small tables, whose
// gradient is all-reduced as a dense buffer across data-parallel ranks).
// (`bid`: the block's index among the blocks of THIS body - the three stepping bodies share one launch, see touched_step_kernel;
//  `s_red`: the launch's 256 doubles of LDS)
template <int LPR>
__device__ __forceinline__ void touched_apply_body(int64_t bid, double* s_red, float4* __restrict__ P, float4* __restrict__ M,
                                                   float4* __restrict__ V, const int32_t* __restrict__ sorted_rows,
                                                   int64_t n, const float4* __restrict__ rowsum,
                                                   const int32_t* __restrict__ head_of, const AdamK& k,
                                                   double* __restrict__ reg_partials, float4* __restrict__ G,
                                                   int32_t* __restrict__ last, int t) {
    const int64_t j = (bid * 256 + threadIdx.x) / LPR;
    const int q = threadIdx.x % LPR;
    double reg = 0.0;
    if (j < n && head_of[j] == 1) {
        const int64_t at = (int64_t)sorted_rows[j] * LPR + q;
        if (G) {
            G[at] = rowsum[j * LPR + q];
        } else {
            float4 p = P[at], m = M[at], v = V[at];
            reg += adam4(p, m, v, rowsum[j * LPR + q], k);
            P[at] = p; M[at] = m; V[at] = v;
            if (last && q == 0) last[sorted_rows[j]] = t;      // lazy form: the row is current as of step t
        }
    }
    const double total = block_sum(reg, s_red);
    if (threadIdx.x == 0) reg_partials[bid] = total;
}

// the runs the superchunk kernel finished, one lane group per chunk (at most one run ends in a chunk that began in an earlier one)
template <int LPR>
__device__ __forceinline__ void touched_finish_body(int64_t bid, double* s_red, float4* __restrict__ P, float4* __restrict__ M,
                                                    float4* __restrict__ V, int64_t chunks,
                                                    const float4* __restrict__ done_sum,
                                                    const int32_t* __restrict__ done_row, const AdamK& k,
                                                    double* __restrict__ reg_partials, float4* __restrict__ G,
                                                    int32_t* __restrict__ last, int t) {
    const int64_t c = (bid * 256 + threadIdx.x) / LPR;
    const int q = threadIdx.x % LPR;
    double reg = 0.0;
    if (c < chunks) {
        const int32_t row = done_row[c];
        if (row >= 0) {
            const int64_t at = (int64_t)row * LPR + q;
            if (G) {
                G[at] = done_sum[c * LPR + q];
            } else {
                float4 p = P[at], m = M[at], v = V[at];
                reg += adam4(p, m, v, done_sum[c * LPR + q], k);
                P[at] = p; M[at] = m; V[at] = v;
                if (last && q == 0) last[row] = t;
            }
        }
    }
    const double total = block_sum(reg, s_red);
    if (threadIdx.x == 0) reg_partials[bid] = total;
}

// third level: one lane group per superchunk that owns a trail piece walks the following superchunks' lead pieces in order
template <int LPR>
__device__ __forceinline__ void touched_spans_body(int64_t bid, double* s_red, float4* __restrict__ P, float4* __restrict__ M,
                                                   float4* __restrict__ V, int64_t chunks,
                                                   const float4* __restrict__ partial,
                                                   const int32_t* __restrict__ info,
                                                   const int32_t* __restrict__ trail_row, const AdamK& k,
                                                   double* __restrict__ reg_partials, float4* __restrict__ G,
                                                   int32_t* __restrict__ last, int t) {
    const int64_t group = (bid * 256 + threadIdx.x) / LPR;
    const int q = threadIdx.x % LPR;
    double reg = 0.0;
    if (group < chunks && (info[group] & 1)) {
        float4 acc = partial[(group * 2 + 1) * LPR + q];
        for (int64_t c = group + 1; c < chunks; ++c) {
            const int f = info[c];
            if (!(f & 2)) break;  // cannot happen for a well-formed trail, kept as a guard
            const float4 g = partial[(c * 2 + 0) * LPR + q];
            acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
            if (f & 4) break;
        }
        const int64_t at = (int64_t)trail_row[group] * LPR + q;
        if (G) {
            G[at] = acc;
        } else {
            float4 p = P[at], m = M[at], v = V[at];
            reg += adam4(p, m, v, acc, k);
            P[at] = p; M[at] = m; V[at] = v;
            if (last && q == 0) last[trail_row[group]] = t;
        }
    }
    const double total = block_sum(reg, s_red);
    if (threadIdx.x == 0) reg_partials[bid] = total;
}

// The three stepping bodies - per-position runs (apply), the runs a superchunk finished (finish), the walk over superchunks
// (spans) - touch disjoint rows and need nothing from each other, only the chunk kernel's and the superchunk kernel's output: ONE
// launch behind the superchunk kernel instead of three in a line.  The few long-running blocks of the walk come first in the grid
// (resident from the start), the many short ones fill the chip around them.  chunks -> super -> [spans | finish | apply]: two
// dependent launches less on the critical path of every step.  (apply merged with the SUPERCHUNK kernel instead, which would make
// the chain a diamond, was measured: that kernel's 80 preloaded registers halve the occupancy the random row reads of apply live
// on - 41 us for the pair against 19 + 16 in a line.)
template <int LPR>
__global__ __launch_bounds__(256) void touched_step_kernel(int64_t sblocks, int64_t cblocks, float4* __restrict__ P,
                                                          float4* __restrict__ M, float4* __restrict__ V, AdamK k,
                                                          float4* __restrict__ G, int32_t* __restrict__ last, int t,
                                                          const int32_t* __restrict__ sorted_rows, int64_t n,
                                                          const float4* __restrict__ rowsum, const int32_t* __restrict__ head_of,
                                                          double* __restrict__ reg_apply, int64_t chunks,
                                                          const float4* __restrict__ done_sum, const int32_t* __restrict__ done_row,
                                                          double* __restrict__ reg_finish, int64_t supers,
                                                          const float4* __restrict__ partial2, const int32_t* __restrict__ info2,
                                                          const int32_t* __restrict__ trail_row2, double* __restrict__ reg_spans) {
    __builtin_amdgcn_s_setprio(3);      // (ahead of the side stream's next-batch kernels when they share a SIMD: 71 -> 62 us for the touched-row chain)
    __shared__ double s_red[256];
    const int64_t b = blockIdx.x;
    if (b < sblocks)
        touched_spans_body<LPR>(b, s_red, P, M, V, supers, partial2, info2, trail_row2, k, reg_spans, G, last, t);
    else if (b < sblocks + cblocks)
        touched_finish_body<LPR>(b - sblocks, s_red, P, M, V, chunks, done_sum, done_row, k, reg_finish, G, last, t);
    else
        touched_apply_body<LPR>(b - sblocks - cblocks, s_red, P, M, V, sorted_rows, n, rowsum, head_of, k, reg_apply, G, last, t);
}

// Dense step over the arena rows [row0, row0 + rows) with a dense gradient buffer G [rows, D] (small tables: every row
// takes g = G + 2*l2*p - for a row nobody gathered G is 0 and that is exactly the regulariser-only step).  last[row] = t.
template <int LPR>
__global__ __launch_bounds__(256) void adam_rows_kernel(float4* __restrict__ P, float4* __restrict__ M,
                                                       float4* __restrict__ V, int64_t row0, int64_t rows,
                                                       const float4* __restrict__ G, int32_t* __restrict__ last, int t,
                                                       AdamK k, double* __restrict__ reg_partials) {
    __shared__ double s_red[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double reg = 0.0;
    if (i < rows * LPR) {
        const int64_t at = row0 * LPR + i;
        float4 p = P[at], m = M[at], v = V[at];
        reg = adam4(p, m, v, G[i], k);
        P[at] = p; M[at] = m; V[at] = v;
        if (last && i % LPR == 0) last[row0 + i / LPR] = t;
    }
    const double total = block_sum(reg, s_red);
    if (threadIdx.x == 0) reg_partials[blockIdx.x] = total;
}

// out[j] = gemb[src[j]]: the gradient rows of a sorted id list in sorted order (what a rank contributes to the all-gather)
template <int LPR>
__global__ void pack_rows_kernel(const int32_t* __restrict__ src, int64_t n, const float4* __restrict__ gemb,
                                 float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * LPR) return;
    out[i] = gemb[(int64_t)src[i / LPR] * LPR + i % LPR];
}


// ---------------------------------------------------------------------------------------------------------
// Lazy-exact form of the dense step.  A row that is not gathered at step s still takes the Adam step of the reference
// (gradient 2*l2*p from the regulariser).  That update depends on nothing but the row's own (p, m, v) and on s (bias
// corrections), so it can be postponed: `last[r]` is the last step applied to row r, and the pending steps
// last[r]+1 .. target are replayed - the same adam1 arithmetic, the per-step constants read from a table the host
// fills exactly as it would fill satrans_adam_hparams - right before the row is next gathered (replay kernel, over
// the run heads of the sorted ids) or for all rows at once (flush kernel: epoch end, before predict / state_dict).
// Bitwise the same tables as the every-step streaming kernel, with HBM traffic only for rows that are touched.
// Each replayed step also contributes l2*|p|^2 (its pre-update value) to the regulariser sum of the epoch.
// ---------------------------------------------------------------------------------------------------------
struct LazyK {
    float w1, beta2, w2, eps, l2x2, l2;
    int fast;
};
__host__ inline LazyK make_lazyk(const satrans_adam_hparams& h) {
    LazyK k;
    k.fast = h.arith == SATRANS_ADAM_FAST ? 1 : 0;
    k.w1 = (float)(1.0 - (double)h.beta1);
    k.beta2 = h.beta2;
    k.w2 = (float)(1.0 - (double)h.beta2);
    k.eps = h.eps;
    k.l2 = h.l2;
    k.l2x2 = 2.0f * h.l2;
    return k;
}

// ---- packed arithmetic of the replay ---------------------------------------------------------------------------------------
// The replay and the flush are bound by VALU issue (one dependent Adam step per element and pending step, ~50 instructions
// with hipcc's IEEE sqrtf and division).  They run four elements per lane as two float2 pairs on the packed fp32 pipe
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per instruction), with the square root and the division
// written out as hipcc's own correctly rounded sequences WITHOUT their range scaling:
//   sqrt: r = rsq(x); s = x r; h = r/2; e = fma(-h, s, 1/2); h = fma(h, e, h); s = fma(s, e, s); s = fma(fma(-s, s, x), h, s)
//   a/b : r = rcp(b); r = fma(fma(-b, r, 1), r, r); q = a r; q = fma(fma(-b, q, a), r, q); q = fma(fma(-b, q, a), r, q)
// The scaling steps (v_div_scale / v_div_fixup, the 2^32 pre-scale of sqrt) only matter when a residual fma would leave the
// exactly representable range; a pair takes the packed path when every operand is inside kPkV / kPkA below, where all
// residuals are exact, and the scalar IEEE path of adam_core otherwise (zero, tiny, huge, NaN).  Both produce the correctly
// rounded result, so the tables stay bit-identical to the streaming kernel: `satrans_debug_check_packed_math` compares the
// sqrt with the fp32 rounding of the fp64 square root over EVERY float of the packed range and the division with the rounded
// fp64 quotient over as many pairs as asked (and adam_core's sqrtf / __fdiv_rn with the same references)
// (tests/test_gpu_parity.py), and test_lazy_adam_is_bitwise_the_streaming_adam stays the end-to-end gate.
using f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ f32x2 pk_set(float x) { return f32x2{x, x}; }

constexpr float kPkVLo = 0x1p-100f, kPkVHi = 0x1p64f;   // exp_avg_sq operands of the packed sqrt
constexpr float kPkALo = 0x1p-80f, kPkAHi = 0x1p40f;    // |step_size * exp_avg| operands of the packed division
constexpr float kPkEpsLo = 0x1p-30f;                    // eps below this: denominators may leave the exact range, scalar kernels only

__device__ __forceinline__ f32x2 pk_sqrt_rn(f32x2 x) {
    const f32x2 r = {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
    f32x2 s = pk_mul(x, r);
    f32x2 h = pk_mul(r, pk_set(0.5f));
    const f32x2 e = pk_fma(-h, s, pk_set(0.5f));
    h = pk_fma(h, e, h);
    s = pk_fma(s, e, s);
    const f32x2 d = pk_fma(-s, s, x);
    return pk_fma(d, h, s);
}

__device__ __forceinline__ f32x2 pk_div_rn(f32x2 a, f32x2 b) {
    f32x2 r = {__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
    const f32x2 e = pk_fma(-b, r, pk_set(1.0f));
    r = pk_fma(e, r, r);
    f32x2 q = pk_mul(a, r);
    f32x2 d = pk_fma(-b, q, a);
    q = pk_fma(d, r, q);
    d = pk_fma(-b, q, a);
    return pk_fma(d, r, q);
}

// ---- the same two operations below the packed range -------------------------------------------------------------------------
// A row nobody gathers decays under the regulariser: |p| and exp_avg fall by a decade every ~60 steps, leave the packed range
// after ~500 steps, pass through the subnormals and end at zero, exp_avg_sq follows over tens of thousands of steps.  These are
// most rows of a long run, and the scalar path costs them 2-3x (measured: the flush of 6.57 M rows went from 9.4 to 15 ms once
// 69 % of the elements had left the range).  Powers of two bring such operands into the packed range and the results back
// exactly:
//   sqrt: v in [2^-149, 2^-100) times 2^64 is in [2^-85, 2^-36); the root comes back by 2^-32 and is never subnormal;
//   a / den: |a| in [2^-149, 2^-80) times 2^69 is in [2^-80, 2^-11); the quotient comes back by 2^-69, exactly as long as it is
//   normal (scaled quotient >= 2^-57) - otherwise `redo` asks the caller for the IEEE division of that element;
//   a = +-0 over a positive denominator is a itself (the sequence would return +0 for -0).
__device__ __forceinline__ f32x2 pk_sqrt_ext(f32x2 v) {
    const bool sx = v.x < kPkVLo, sy = v.y < kPkVLo;
    const f32x2 up = {sx ? 0x1p64f : 1.0f, sy ? 0x1p64f : 1.0f};
    const f32x2 dn = {sx ? 0x1p-32f : 1.0f, sy ? 0x1p-32f : 1.0f};
    return pk_mul(pk_sqrt_rn(pk_mul(v, up)), dn);
}

__device__ __forceinline__ f32x2 pk_div_ext(f32x2 a, f32x2 den, bool& redo_x, bool& redo_y) {
    const bool sx = fabsf(a.x) < kPkALo, sy = fabsf(a.y) < kPkALo;
    const f32x2 up = {sx ? 0x1p69f : 1.0f, sy ? 0x1p69f : 1.0f};
    const f32x2 dn = {sx ? 0x1p-69f : 1.0f, sy ? 0x1p-69f : 1.0f};
    const f32x2 qs = pk_div_rn(pk_mul(a, up), den);
    f32x2 q = pk_mul(qs, dn);
    redo_x = sx && a.x != 0.f && !(fabsf(qs.x) >= 0x1p-57f);
    redo_y = sy && a.y != 0.f && !(fabsf(qs.y) >= 0x1p-57f);
    q.x = a.x == 0.f ? a.x : q.x;
    q.y = a.y == 0.f ? a.y : q.y;
    return q;
}

// sqrt(v) / bc2_sqrt + eps of adam_core for one element (the division by the per-step constant, see there)
__device__ __forceinline__ float denom_of(float root, double rbc2, float eps) {
    return __fadd_rn((float)__dmul_rn((double)root, rbc2), eps);
}

// One regulariser-only Adam step of four elements; bit for bit four calls of adam_core with g = 0 + 2*l2*p.
__device__ __forceinline__ void adam_step4(f32x2 (&p)[2], f32x2 (&m)[2], f32x2 (&v)[2], f32x2 (&sq)[2], float neg_step,
                                           double rbc2, const LazyK& k) {
    f32x2 a[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        sq[u] = pk_fma(p[u], p[u], sq[u]);
        const f32x2 g = pk_add(pk_set(0.f), pk_mul(pk_set(k.l2x2), p[u]));
        m[u] = pk_fma(pk_set(k.w1), pk_sub(g, m[u]), m[u]);
        v[u] = pk_fma(pk_mul(pk_set(k.w2), g), g, pk_mul(v[u], pk_set(k.beta2)));
        a[u] = pk_mul(pk_set(neg_step), m[u]);
    }
    const float vmin = fminf(__builtin_fminf(__builtin_fminf(v[0].x, v[0].y), v[1].x), v[1].y);
    const float vmax = fmaxf(__builtin_fmaxf(__builtin_fmaxf(v[0].x, v[0].y), v[1].x), v[1].y);
    const float amin = fminf(__builtin_fminf(__builtin_fminf(fabsf(a[0].x), fabsf(a[0].y)), fabsf(a[1].x)), fabsf(a[1].y));
    const float amax = fmaxf(__builtin_fmaxf(__builtin_fmaxf(fabsf(a[0].x), fabsf(a[0].y)), fabsf(a[1].x)), fabsf(a[1].y));
    // NaNs drop out of min / max, and a NaN operand gives NaN on either path
    const bool packed = vmin >= kPkVLo && vmax <= kPkVHi && amin >= kPkALo && amax <= kPkAHi && k.eps >= kPkEpsLo;
    // The path is chosen per WAVE (a vote), so that a wave runs one of them:
    //  1. the packed sequences when every element of every lane is inside their range (196 ns per step and SIMD);
    //  2. the same with ONE power of two per lane on the numerators (a row decays as a whole: its four elements of a lane are
    //     scaled by 2^69 together when the largest is below 2^-60) and zero numerators passed through - rows that are decaying
    //     or have decayed to zero, mixed with live ones (~1.2x);
    //  3. the element-wise scaling of pk_sqrt_ext / pk_div_ext when every element is at least inside the extended range (second
    //     moments down to the smallest subnormal, lanes whose elements are far apart) (~1.5x);
    //  4. adam_core's IEEE operations otherwise (huge values, NaN, a zero second moment under a non-zero numerator, a tiny eps).
    if (__all(packed)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x2 root = pk_sqrt_rn(v[u]);
            const f32x2 den = {denom_of(root.x, rbc2, k.eps), denom_of(root.y, rbc2, k.eps)};
            p[u] = pk_add(p[u], pk_div_rn(a[u], den));
        }
        return;
    }
    // (0. a wave of rows that have decayed to zero: p + (+-0) is all there is to compute)
    // (only over a POSITIVE denominator is +-0 / den = +-0: eps = 0 with a zero second moment is 0 / 0 = NaN in adam_core and in
    //  torch, and a NaN second moment is a NaN there - `v >= 0` is false for a NaN; both fall through to the IEEE tier)
    const bool v_sane = v[0].x >= 0.f && v[0].y >= 0.f && v[1].x >= 0.f && v[1].y >= 0.f;
    if (__all(k.eps > 0.f && v_sane && a[0].x == 0.f && a[0].y == 0.f && a[1].x == 0.f && a[1].y == 0.f)) {
        p[0] = pk_add(p[0], a[0]);
        p[1] = pk_add(p[1], a[1]);
        return;
    }
    const bool lane_small = amax < 0x1p-60f;
    auto lane_ok = [&](float a_) { return lane_small || a_ == 0.f || fabsf(a_) >= kPkALo; };
    const bool by_lane = v_sane && vmin >= kPkVLo && vmax <= kPkVHi && amax <= kPkAHi && k.eps >= kPkEpsLo && lane_ok(a[0].x) &&
                         lane_ok(a[0].y) && lane_ok(a[1].x) && lane_ok(a[1].y);
    if (__all(by_lane)) {
        const f32x2 up = pk_set(lane_small ? 0x1p69f : 1.0f), dn = pk_set(lane_small ? 0x1p-69f : 1.0f);
        f32x2 q[2], den[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x2 root = pk_sqrt_rn(v[u]);
            den[u] = f32x2{denom_of(root.x, rbc2, k.eps), denom_of(root.y, rbc2, k.eps)};
            q[u] = pk_mul(pk_div_rn(pk_mul(a[u], up), den[u]), dn);
        }
        // the way back is exact while the quotient is normal: scaled numerators are >= 2^-80, so denominators up to 2^-23
        // (eps + a small root: the decayed rows) are safe; a scaled lane with a larger one divides in IEEE
        if (lane_small && fmaxf(fmaxf(den[0].x, den[0].y), fmaxf(den[1].x, den[1].y)) > 0x1p-23f) {
#pragma unroll
            for (int u = 0; u < 2; ++u) q[u] = f32x2{__fdiv_rn(a[u].x, den[u].x), __fdiv_rn(a[u].y, den[u].y)};
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            q[u].x = a[u].x == 0.f ? a[u].x : q[u].x;        // (+-0 over a positive denominator is itself; the sequence returns +0)
            q[u].y = a[u].y == 0.f ? a[u].y : q[u].y;
            p[u] = pk_add(p[u], q[u]);
        }
        return;
    }
    auto ext_ok = [&](float a_, float v_) { return (a_ == 0.f && v_ >= 0.f) || (fabsf(a_) <= kPkAHi && v_ > 0.f && v_ <= kPkVHi); };
    const bool extended = ext_ok(a[0].x, v[0].x) && ext_ok(a[0].y, v[0].y) && ext_ok(a[1].x, v[1].x) && ext_ok(a[1].y, v[1].y) &&
                          k.eps >= kPkEpsLo;
    if (__all(extended)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x2 root = pk_sqrt_ext(v[u]);
            const f32x2 den = {denom_of(root.x, rbc2, k.eps), denom_of(root.y, rbc2, k.eps)};
            bool redo_x, redo_y;
            f32x2 q = pk_div_ext(a[u], den, redo_x, redo_y);
            if (redo_x) q.x = __fdiv_rn(a[u].x, den.x);
            if (redo_y) q.y = __fdiv_rn(a[u].y, den.y);
            p[u] = pk_add(p[u], q);
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        p[u].x = __fadd_rn(p[u].x, __fdiv_rn(a[u].x, denom_of(sqrtf(v[u].x), rbc2, k.eps)));
        p[u].y = __fadd_rn(p[u].y, __fdiv_rn(a[u].y, denom_of(sqrtf(v[u].y), rbc2, k.eps)));
    }
}

// The same step in the FAST arithmetic (adam_core_fast, four elements): one path for every value - no range votes, no scaling.
__device__ __forceinline__ void adam_step4_fast(f32x2 (&p)[2], f32x2 (&m)[2], f32x2 (&v)[2], f32x2 (&sq)[2], float neg_step,
                                                float rbc2f, const LazyK& k) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        sq[u] = pk_fma(p[u], p[u], sq[u]);
        const f32x2 g = pk_add(pk_set(0.f), pk_mul(pk_set(k.l2x2), p[u]));
        m[u] = pk_fma(pk_set(k.w1), pk_sub(g, m[u]), m[u]);
        v[u] = pk_fma(pk_mul(pk_set(k.w2), g), g, pk_mul(v[u], pk_set(k.beta2)));
        const f32x2 a = pk_mul(pk_set(neg_step), m[u]);
        const f32x2 root = {__builtin_amdgcn_sqrtf(v[u].x), __builtin_amdgcn_sqrtf(v[u].y)};
        const f32x2 den = pk_fma(root, pk_set(rbc2f), pk_set(k.eps));
        const f32x2 r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
        p[u] = pk_fma(a, r, p[u]);
    }
}

// four elements per lane; steps (from, to] ; table[s] = (lr / (1 - beta1^s) as fp32, 1 / (double)fp32(sqrt(1 - beta2^s))) as doubles
__device__ __forceinline__ double replay_element4(float4& P4, float4& M4, float4& V4, int from, int to,
                                                  const double2* __restrict__ table, const LazyK& k) {
    // sum of p^2 over the replayed steps in fp32 (<= a few thousand terms of one element: relative error ~1e-7 x steps,
    // far inside the fp32 reduction the reference itself uses), converted to double once; the sums over elements,
    // blocks and steps stay in double
    f32x2 p[2] = {{P4.x, P4.y}, {P4.z, P4.w}}, m[2] = {{M4.x, M4.y}, {M4.z, M4.w}}, v[2] = {{V4.x, V4.y}, {V4.z, V4.w}};
    f32x2 sq[2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (k.fast) {
        for (int s = from + 1; s <= to; ++s) {
            const double2 hp = table[s];
            adam_step4_fast(p, m, v, sq, -(float)hp.x, (float)hp.y, k);
        }
    } else {
        for (int s = from + 1; s <= to; ++s) {
            const double2 hp = table[s];
            adam_step4(p, m, v, sq, -(float)hp.x, hp.y, k);
        }
    }
    P4 = make_float4(p[0].x, p[0].y, p[1].x, p[1].y);
    M4 = make_float4(m[0].x, m[0].y, m[1].x, m[1].y);
    V4 = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
    return (double)k.l2 * (((double)sq[0].x + (double)sq[0].y) + ((double)sq[1].x + (double)sq[1].y));
}

// replay for the rows of one batch: one group of D/4 lanes per sorted position, run heads only.  `slots` partial sums are
// written (the workspace is sized for one per 256 elements): this grid's one per block, zeros in the rest.
template <int D>
__global__ __launch_bounds__(256) void lazy_replay_kernel(float4* __restrict__ P, float4* __restrict__ M, float4* __restrict__ V,
                                                         int32_t* __restrict__ last, const int32_t* __restrict__ sorted_rows,
                                                         int64_t n, int target, const double2* __restrict__ table, LazyK k,
                                                         double* __restrict__ reg_partials, int64_t slots) {
    constexpr int LR = D / 4;
    __shared__ double s_red[256];
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LR;
    const int c = threadIdx.x % LR;
    double reg = 0.0;
    if (j < n) {
        const int32_t row = sorted_rows[j];
        const bool head = j == 0 || sorted_rows[j - 1] != row;
        if (head) {
            const int from = last[row];
            if (from < target) {
                const int64_t at = (int64_t)row * LR + c;
                float4 p = P[at], m = M[at], v = V[at];
                reg = replay_element4(p, m, v, from, target, table, k);
                P[at] = p; M[at] = m; V[at] = v;
            }
        }
    }
    // every lane of a row has read last[row] before any lane overwrites it
    __syncthreads();
    if (j < n && c == 0) {
        const int32_t row = sorted_rows[j];
        if ((j == 0 || sorted_rows[j - 1] != row) && last[row] < target) last[row] = target;
    }
    const double total = block_sum(reg, s_red);
    if (threadIdx.x == 0) {
        reg_partials[blockIdx.x] = total;
        for (int64_t t = (int64_t)blockIdx.x + gridDim.x; t < slots; t += gridDim.x) reg_partials[t] = 0.0;
    }
}

// replay for every row of the arena
template <int D>
__global__ __launch_bounds__(256) void lazy_flush_kernel(float4* __restrict__ P, float4* __restrict__ M, float4* __restrict__ V,
                                                        int32_t* __restrict__ last, int64_t total_rows, int target,
                                                        const double2* __restrict__ table, LazyK k,
                                                        double* __restrict__ reg_partials) {
    constexpr int LR = D / 4;
    __shared__ double s_red[256];
    double reg = 0.0;
    const int64_t groups_per_pass = (int64_t)gridDim.x * 256 / LR;
    const int c = threadIdx.x % LR;
    for (int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LR; row < total_rows; row += groups_per_pass) {
        const int from = last[row];
        if (from < target) {
            const int64_t at = row * LR + c;
            float4 p = P[at], m = M[at], v = V[at];
            reg += replay_element4(p, m, v, from, target, table, k);
            P[at] = p; M[at] = m; V[at] = v;
        }
    }
    __syncthreads();    // all reads of last[] in this block are done; blocks own disjoint rows
    for (int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LR; row < total_rows; row += groups_per_pass)
        if (c == 0 && last[row] < target) last[row] = target;
    const double total = block_sum(reg, s_red);
    if (threadIdx.x == 0) reg_partials[blockIdx.x] = total;
}

// Diagnostic behind the packed arithmetic: mode 0 compares pk_sqrt_rn with RN32(sqrt in fp64) on the floats whose bit patterns are
// [first, first + count) (the caller walks the whole packed range), mode 1 compares pk_div_rn with RN32(fp64 quotient) on `count`
// pseudo-random (a, b) pairs drawn from seed `first` with a in +-[2^-80, 2^40], b in [2^-30, 2^38]; mismatches are counted.
__global__ void packed_math_check_kernel(int mode, uint64_t first, uint64_t count, unsigned long long* __restrict__ bad) {
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        if (mode == 2) {
            // the scaled square root: every positive float below the packed range, subnormals included
            const float x = __uint_as_float((uint32_t)(first + i));
            if (!(x > 0.f && x < kPkVLo)) continue;
            const f32x2 r = pk_sqrt_ext(f32x2{x, x});
            const float want = (float)sqrt((double)x);
            mine += (__float_as_uint(r.x) != __float_as_uint(want)) + (__float_as_uint(r.y) != __float_as_uint(want));
        } else if (mode == 3) {
            // the scaled division: numerators of both signs from the smallest subnormal to 2^-80 (exponent field uniform over
            // 0 .. 47, a zero field being the subnormals), denominators over the packed range; the elements it hands back
            // (`redo`) take the IEEE division as in adam_step4
            uint64_t z = (first + i) * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            uint64_t w = (z + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
            w ^= w >> 29;
            const uint32_t ea = (uint32_t)((w >> 8) % 48u);
            const uint32_t eb = 127 - 30 + (uint32_t)((w >> 20) % 68u);
            float a = __uint_as_float(((uint32_t)(w & 1u) << 31) | (ea << 23) | (uint32_t)(z & 0x7FFFFFu));
            if ((w >> 40) % 64u == 0) a = __uint_as_float((uint32_t)(w & 1u) << 31);      // (signed zeros now and then)
            const float b = __uint_as_float((eb << 23) | (uint32_t)((z >> 23) & 0x7FFFFFu));
            bool rx, ry;
            f32x2 q = pk_div_ext(f32x2{a, -a}, f32x2{b, b}, rx, ry);
            if (rx) q.x = __fdiv_rn(a, b);
            if (ry) q.y = __fdiv_rn(-a, b);
            const float want = (float)((double)a / (double)b);
            mine += __float_as_uint(__fdiv_rn(a, b)) != __float_as_uint(want);
            mine += (__float_as_uint(q.x) != __float_as_uint(want)) + (__float_as_uint(q.y) != __float_as_uint(-want));
        } else if (mode == 0) {
            const float x = __uint_as_float((uint32_t)(first + i));
            if (!(x >= kPkVLo && x <= kPkVHi)) continue;
            const f32x2 r = pk_sqrt_rn(f32x2{x, x});
            const float want = (float)sqrt((double)x);       // correctly rounded: 53 >= 2*24 + 2 bits make the double rounding innocuous
            mine += __float_as_uint(sqrtf(x)) != __float_as_uint(want);   // ... and hipcc's own fp32 sqrt (adam_core's) agrees
            mine += (__float_as_uint(r.x) != __float_as_uint(want)) + (__float_as_uint(r.y) != __float_as_uint(want));
        } else {
            // splitmix64 of the pair index: mantissas uniform, exponents uniform over the packed range
            uint64_t z = (first + i) * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            uint64_t w = (z + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
            w ^= w >> 29;
            const uint32_t ea = 127 - 80 + (uint32_t)((w >> 8) % 120u);        // 2^-80 .. 2^39
            const uint32_t eb = 127 - 30 + (uint32_t)((w >> 20) % 68u);        // 2^-30 .. 2^37
            const float a = __uint_as_float(((uint32_t)(w & 1u) << 31) | (ea << 23) | (uint32_t)(z & 0x7FFFFFu));
            const float b = __uint_as_float((eb << 23) | (uint32_t)((z >> 23) & 0x7FFFFFu));
            const f32x2 q = pk_div_rn(f32x2{a, -a}, f32x2{b, b});
            const float want = (float)((double)a / (double)b);   // correctly rounded for the same reason
            mine += __float_as_uint(__fdiv_rn(a, b)) != __float_as_uint(want);
            mine += (__float_as_uint(q.x) != __float_as_uint(want)) + (__float_as_uint(q.y) != __float_as_uint(-want));
        }
    }
    if (mine) atomicAdd(bad, mine);
}

__global__ void mark_last_kernel(const int32_t* __restrict__ sorted_rows, int64_t n, int32_t* __restrict__ last, int t) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t row = sorted_rows[j];
    if (j == 0 || sorted_rows[j - 1] != row) last[row] = t;
}

// debug / parity: dense gradient of the arena
template <int LPR>
__global__ void grad_dense_kernel(const int32_t* __restrict__ sorted_rows, const int32_t* __restrict__ src, int64_t n,
                                  const float4* __restrict__ gemb, float4* __restrict__ g_arena) {
    const int64_t j0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
    const int q = threadIdx.x % LPR;
    if (j0 >= n) return;
    const int32_t row = sorted_rows[j0];
    if (j0 > 0 && sorted_rows[j0 - 1] == row) return;  // only segment heads work
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t j = j0; j < n && sorted_rows[j] == row; ++j) {
        const float4 g = gemb[(int64_t)src[j] * LPR + q];
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
    }
    float4 out = g_arena[(int64_t)row * LPR + q];
    out.x += acc.x; out.y += acc.y; out.z += acc.z; out.w += acc.w;
    g_arena[(int64_t)row * LPR + q] = out;
}

__global__ void add_l2_grad_kernel(const float* __restrict__ p, float* __restrict__ g, int64_t n, float l2x2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] += l2x2 * p[i];
}

__global__ void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, int64_t n, AdamK k) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float pp = p[i], mm = m[i], vv = v[i];
    adam1(pp, mm, vv, __fadd_rn(g[i], __fmul_rn(k.l2x2, pp)), k);
    p[i] = pp; m[i] = mm; v[i] = vv;
}

// vals[threadIdx.x], vals[threadIdx.x + 1024], ... added in index order (a workgroup of 1024 lanes); sixteen loads in flight per
// lane - the same bits as one load at a time, without paying the memory latency once per element (13.5 -> 10 us with eight in
// flight for the ~40 k partial sums of a step)
__device__ __forceinline__ double strided_sum_1024(const double* __restrict__ vals, int64_t count) {
    constexpr int kDeep = 16;
    double acc = 0.0;
    int64_t i = threadIdx.x;
    for (; i + (kDeep - 1) * 1024 < count; i += kDeep * 1024) {
        double t[kDeep];
#pragma unroll
        for (int u = 0; u < kDeep; ++u) t[u] = vals[i + u * 1024];
#pragma unroll
        for (int u = 0; u < kDeep; ++u) acc += t[u];
    }
    if (i < count) {      // the rest: still all loads first (clamped index, masked sum)
        double t[kDeep];
#pragma unroll
        for (int u = 0; u < kDeep; ++u) t[u] = vals[min(i + (int64_t)u * 1024, count - 1)];
#pragma unroll
        for (int u = 0; u < kDeep; ++u)
            if (i + (int64_t)u * 1024 < count) acc += t[u];
    }
    return acc;
}

// flat Adam and the step's regulariser sum in one launch: the last block adds the partial sums in a fixed order
__global__ __launch_bounds__(1024) void adam_flat_sum_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, int64_t n, AdamK k,
                                                           const double* __restrict__ vals, int64_t count,
                                                           double* __restrict__ out) {
    __builtin_amdgcn_s_setprio(3);      // (ahead of the side stream's next-batch kernels when they share a SIMD: 71 -> 62 us for the touched-row chain)
    __shared__ double s_red[1024];
    if (blockIdx.x + 1 < gridDim.x) {
        const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n) return;
        float pp = p[i], mm = m[i], vv = v[i];
        adam1(pp, mm, vv, __fadd_rn(g[i], __fmul_rn(k.l2x2, pp)), k);
        p[i] = pp; m[i] = mm; v[i] = vv;
        return;
    }
    const double total = block_sum(strided_sum_1024(vals, count), s_red);
    if (threadIdx.x == 0) out[0] += total;
}

__global__ __launch_bounds__(1024) void sum_f64_kernel(const double* __restrict__ vals, int64_t count,
                                                      double* __restrict__ out, int accumulate) {
    __shared__ double s_red[1024];
    const double acc = strided_sum_1024(vals, count);
    const double total = block_sum(acc, s_red);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + total : total;
}

static int bits_for(int64_t n) {
    int bits = 1;
    while (((int64_t)1 << bits) < n) ++bits;
    return bits;
}

struct SortLayout {
    size_t keys_in, pos_in, temp, temp_bytes, total;
};
static SortLayout sort_layout(int64_t n, int64_t total_rows) {
    SortLayout L{};
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t temp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, temp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
                              (int32_t*)nullptr, (unsigned)n, 0u, (unsigned)bits_for(total_rows), (hipStream_t)0);
    L.keys_in = 0;
    L.pos_in = align(sizeof(uint32_t) * (size_t)n);
    L.temp = align(L.pos_in + sizeof(int32_t) * (size_t)n);
    L.temp_bytes = temp;
    L.total = align(L.temp + temp);
    return L;
}

static int64_t touched_blocks(int64_t n, int D) {   // blocks of the per-position apply kernel (the larger of the two)
    const int lpr = D / 4;
    return ceil_div(n * lpr, 256);
}
static int64_t chunk_blocks(int64_t n, int D) {
    const int lpr = D / 4;
    return ceil_div(ceil_div(n, kChunk) * lpr, 256);
}

}  // namespace satrans

using namespace satrans;

#define DISPATCH_LPR(D, CALL)                                                        \
    switch ((D) / 4) {                                                               \
        case 4: { constexpr int LPR = 4; CALL; } break;                              \
        case 8: { constexpr int LPR = 8; CALL; } break;                              \
        case 16: { constexpr int LPR = 16; CALL; } break;                            \
        case 32: { constexpr int LPR = 32; CALL; } break;                            \
        default:                                                                     \
            ::satrans::set_error("embedding_dim %d not in {16,32,64,128}", (D));     \
            return SATRANS_E_UNSUPPORTED;                                            \
    }

extern "C" int64_t satrans_embed_sort_workspace_bytes(int64_t n, int64_t total_rows) {
    if (n <= 0 || total_rows <= 0) return 0;
    return (int64_t)sort_layout(n, total_rows).total;
}

extern "C" int satrans_embed_sort(const int32_t* rows, int64_t n, int64_t total_rows, int32_t* sorted_rows, int32_t* src,
                                  uint32_t* touched, void* workspace, int64_t workspace_bytes, const int32_t* positions,
                                  void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(rows && sorted_rows && src && workspace, SATRANS_E_BADARG, "embed_sort: null pointer");
    SATRANS_REQUIRE(n > 0 && total_rows > 0 && n < ((int64_t)1 << 31) && total_rows < ((int64_t)1 << 31), SATRANS_E_BADARG,
                    "embed_sort: sizes n=%lld rows=%lld", (long long)n, (long long)total_rows);
    const SortLayout L = sort_layout(n, total_rows);
    SATRANS_REQUIRE((int64_t)L.total <= workspace_bytes, SATRANS_E_WORKSPACE, "embed_sort: workspace %lld < %lld bytes",
                    (long long)workspace_bytes, (long long)L.total);
    char* ws = (char*)workspace;
    uint32_t* keys_in = (uint32_t*)(ws + L.keys_in);
    int32_t* pos_in = (int32_t*)(ws + L.pos_in);
    hipError_t e = hipSuccess;
    if (touched) e = hipMemsetAsync(touched, 0, sizeof(uint32_t) * (size_t)ceil_div(total_rows, 32), stream);
    SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "embed_sort: memset: %s", hipGetErrorString(e));
    const uint32_t* keys = keys_in;
    const int32_t* pos = pos_in;
    if (positions) {     // the caller keeps positions[i] = i around: the rows are the keys as they are, no preparation launch
        keys = (const uint32_t*)rows;
        pos = positions;
    } else {
        iota_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(rows, n, keys_in, pos_in);
        SATRANS_CHECK_LAUNCH("iota_kernel");
    }
    size_t temp_bytes = L.temp_bytes;
    e = rocprim::radix_sort_pairs(ws + L.temp, temp_bytes, keys, (uint32_t*)sorted_rows, pos, src, (unsigned)n, 0u,
                                  (unsigned)bits_for(total_rows), stream);
    SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "embed_sort: radix sort: %s", hipGetErrorString(e));
    if (touched) {   // only the every-step streaming Adam needs the bitmap
        mark_heads_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(sorted_rows, n, touched);
        SATRANS_CHECK_LAUNCH("mark_heads_kernel");
    }
    return SATRANS_OK;
}

// ---- W sorted runs back to back -> one sorted list (the owner form: what W ranks sent an owner is W sorted runs) ---------------------
// Output identical to satrans_embed_sort(ids, positions = 0..n-1): ascending rows, equal rows in position order - positions grow
// with (run, index inside the run), and a run is itself sorted that way.  One lane per element: its final position is its index
// inside its run + for every EARLIER run the number of elements <= it + for every LATER run the number of elements < it (W - 1
// binary searches over L2-resident runs, eight of them in flight at a time) - one launch instead of the device-wide sort's block
// sort + ~10 merge passes.
constexpr int kMergeRunsMax = 64;
struct MergeRuns {
    int32_t start[kMergeRunsMax + 1];
};
__global__ __launch_bounds__(256) void merge_runs_kernel(const int32_t* __restrict__ ids, int n, int W, MergeRuns R,
                                                         int32_t* __restrict__ sorted_rows, int32_t* __restrict__ src) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t x = ids[i];
    int r = 0;
    while (r + 1 < W && i >= R.start[r + 1]) ++r;      // (W <= 64: a short scan over kernel arguments)
    int pos = i - R.start[r];
    for (int q0 = 0; q0 < W; q0 += 8) {
        int lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = q0 + k;
            const bool on = q < W && q != r;
            lo[k] = on ? R.start[q] : 0;
            hi[k] = on ? R.start[q + 1] : 0;
        }
        bool more = true;
        while (more) {
            more = false;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (lo[k] < hi[k]) {
                    const int mid = (lo[k] + hi[k]) >> 1;
                    const int32_t v = ids[mid];
                    // earlier run: count elements <= x (upper bound); later run: elements < x (lower bound)
                    const bool right = (q0 + k < r) ? (v <= x) : (v < x);
                    if (right) lo[k] = mid + 1; else hi[k] = mid;
                    more = true;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = q0 + k;
            if (q < W && q != r) pos += lo[k] - R.start[q];
        }
    }
    sorted_rows[pos] = x;
    src[pos] = i;
}

// h_run_start: HOST array of W + 1 ints, run q = ids[h_run_start[q], h_run_start[q + 1]); W <= 64
extern "C" int satrans_embed_merge_runs(const int32_t* ids, int64_t n, const int64_t* h_run_start, int W, int32_t* sorted_rows,
                                        int32_t* src, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(ids && h_run_start && sorted_rows && src, SATRANS_E_BADARG, "embed_merge_runs: null pointer");
    SATRANS_REQUIRE(W >= 1 && W <= kMergeRunsMax, SATRANS_E_UNSUPPORTED, "embed_merge_runs: %d runs (1..%d)", W, kMergeRunsMax);
    SATRANS_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && h_run_start[0] == 0 && h_run_start[W] == n, SATRANS_E_BADARG,
                    "embed_merge_runs: run boundaries do not cover [0, %lld)", (long long)n);
    MergeRuns R;
    for (int q = 0; q <= kMergeRunsMax; ++q) R.start[q] = (int32_t)h_run_start[std::min(q, W)];
    for (int q = 0; q < W; ++q)
        SATRANS_REQUIRE(R.start[q] <= R.start[q + 1], SATRANS_E_BADARG, "embed_merge_runs: run %d has a negative length", q);
    if (n == 0) return SATRANS_OK;
    merge_runs_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(ids, (int)n, W, R, sorted_rows, src);
    SATRANS_CHECK_LAUNCH("merge_runs_kernel");
    return SATRANS_OK;
}

// inv[src[i]] = i: the inverse of a sort's source positions (token (b, f) -> its place in the sorted list)
__global__ __launch_bounds__(256) void inverse_positions_kernel(const int32_t* __restrict__ src, int n, int32_t* __restrict__ inv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) inv[src[i]] = i;
}
extern "C" int satrans_embed_inverse_positions(const int32_t* src, int64_t n, int32_t* inv, void* stream_) {
    SATRANS_REQUIRE(src && inv && n >= 0 && n < ((int64_t)1 << 31), SATRANS_E_BADARG, "embed_inverse_positions: bad argument");
    if (n == 0) return SATRANS_OK;
    inverse_positions_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, (hipStream_t)stream_>>>(src, (int)n, inv);
    SATRANS_CHECK_LAUNCH("inverse_positions_kernel");
    return SATRANS_OK;
}

// ---- the same sort for the rows of ONE batch when every field has a table of its own ---------------------------------------------
// rows [B, F] (position = b F + f).  The tables lie back to back in the arena, so the globally sorted list is the fields' own
// sorted lists in arena order: one workgroup per field sorts its B (row, position) pairs in LDS (rocPRIM block radix sort, stable,
// only the bits that field's table needs) - one launch where the device-wide sort runs a block sort and ~10 merge passes of
// 5-6 us each for these 10^5 keys.  Output identical to satrans_embed_sort (ascending rows, equal rows in position order).
#ifndef SATRANS_SORT_RADIX_BITS
#define SATRANS_SORT_RADIX_BITS 8      // key bits per pass of the in-LDS sort
#endif
constexpr int kSortFieldsMax = 64;
struct SortFields {
    int32_t field[kSortFieldsMax];   // field index of the k-th segment (arena order)
    int32_t lo[kSortFieldsMax];      // first arena row of that field's table
    int32_t bits[kSortFieldsMax];    // key bits of its local ids
};

// IDS: the keys come straight from the id matrix (satrans_embed_rows_sort_fields) - ids -> arena rows exactly as the gather kernel
// translates them (gather.hip: an id outside its table is flagged in `status` and recorded as the table's first row) - and the
// [B, F] row matrix is written on the way: the launch that used to produce it is gone from the front of every sort.
struct SortIds {
    const void* X;
    int id_dtype;
    int64_t x_stride;
    const int32_t* cols;
    const int64_t* row_span;
    int32_t* rows_out;
    int32_t* status;
};

template <int ITEMS, bool IDS = false>
__global__ __launch_bounds__(1024) void sort_fields_kernel(const int32_t* __restrict__ rows, int B, int F, SortFields sf,
                                                          int32_t* __restrict__ sorted_rows, int32_t* __restrict__ src,
                                                          SortIds ids) {
    using Sort = rocprim::block_radix_sort<uint32_t, 1024, ITEMS, int32_t, 1, 1, SATRANS_SORT_RADIX_BITS>;
    __shared__ typename Sort::storage_type storage;
    const int seg = blockIdx.x, f = sf.field[seg], lo = sf.lo[seg];
    uint32_t keys[ITEMS];
    int32_t vals[ITEMS];
    int col = 0;
    int64_t span = 0;
    if constexpr (IDS) {
        col = ids.cols[f];
        span = ids.row_span[2 * f + 1] - ids.row_span[2 * f];      // (row_span[2 f] == lo: checked by the launcher's caller contract)
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int b = (int)threadIdx.x * ITEMS + i;           // blocked arrangement: sample order = position order
        const bool in = b < B;
        if constexpr (IDS) {
            uint32_t key = 0xFFFFFFFFu;
            if (in) {
                const int64_t id = load_id(ids.X, ids.id_dtype, ids.x_stride, b, col);
                const bool bad = id < 0 || id >= span;
                key = bad ? 0u : (uint32_t)id;
                ids.rows_out[(size_t)b * F + f] = lo + (int32_t)key;
                if (bad) atomicOr(ids.status, 1);
            }
            keys[i] = key;
        } else {
            keys[i] = in ? (uint32_t)(rows[(size_t)b * F + f] - lo) : 0xFFFFFFFFu;      // padding sorts to the end
        }
        vals[i] = in ? b * F + f : -1;
    }
    // padding keys need the top bit: sort one bit more than the ids use when the batch does not fill the block
    const unsigned end_bit = B < 1024 * ITEMS ? 32u : (unsigned)sf.bits[seg];
    Sort().sort(keys, vals, storage, 0u, end_bit);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int j = (int)threadIdx.x * ITEMS + i;
        if (j < B) {
            sorted_rows[(size_t)seg * B + j] = (int32_t)keys[i] + lo;
            src[(size_t)seg * B + j] = vals[i];
        }
    }
}

// seg_field / seg_lo / seg_rows: HOST arrays of F entries, the fields in arena order with the first row and the row count of their
// tables (pairwise disjoint).  B <= 8192, F <= 64; returns SATRANS_E_UNSUPPORTED otherwise (use satrans_embed_sort).
extern "C" int satrans_embed_sort_fields(const int32_t* rows, int B, int F, const int32_t* seg_field, const int32_t* seg_lo,
                                         const int32_t* seg_rows, int32_t* sorted_rows, int32_t* src, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(rows && seg_field && seg_lo && seg_rows && sorted_rows && src, SATRANS_E_BADARG, "embed_sort_fields: null pointer");
    SATRANS_REQUIRE(B > 0 && B <= 8192 && F > 0 && F <= kSortFieldsMax, SATRANS_E_UNSUPPORTED,
                    "embed_sort_fields: B=%d F=%d (B <= 8192, F <= %d)", B, F, kSortFieldsMax);
    SortFields sf;
    for (int k = 0; k < F; ++k) {
        SATRANS_REQUIRE(seg_field[k] >= 0 && seg_field[k] < F && seg_rows[k] > 0 && (k == 0 || seg_lo[k] >= seg_lo[k - 1] + seg_rows[k - 1]),
                        SATRANS_E_BADARG, "embed_sort_fields: segment %d is not a table of its own behind segment %d", k, k - 1);
        sf.field[k] = seg_field[k];
        sf.lo[k] = seg_lo[k];
        sf.bits[k] = bits_for(seg_rows[k]);
    }
    for (int k = F; k < kSortFieldsMax; ++k) { sf.field[k] = 0; sf.lo[k] = 0; sf.bits[k] = 1; }
    const SortIds none{};
    if (B <= 1024) sort_fields_kernel<1><<<F, 1024, 0, stream>>>(rows, B, F, sf, sorted_rows, src, none);
    else if (B <= 2048) sort_fields_kernel<2><<<F, 1024, 0, stream>>>(rows, B, F, sf, sorted_rows, src, none);
    else if (B <= 4096) sort_fields_kernel<4><<<F, 1024, 0, stream>>>(rows, B, F, sf, sorted_rows, src, none);
    else sort_fields_kernel<8><<<F, 1024, 0, stream>>>(rows, B, F, sf, sorted_rows, src, none);
    SATRANS_CHECK_LAUNCH("sort_fields_kernel");
    return SATRANS_OK;
}

// satrans_gather_fwd(out = NULL, rows_out = rows) + satrans_embed_sort_fields(rows, ...) in one launch: `rows` [B, F] is an OUTPUT.
// row_span / cols: the device arrays satrans_gather_fwd takes; seg_lo[k] must be row_span[2 seg_field[k]] (the field's own table).
extern "C" int satrans_embed_rows_sort_fields(const void* X, int id_dtype, int64_t x_stride, const int32_t* cols,
                                              const int64_t* row_span, int32_t* rows, int B, int F, const int32_t* seg_field,
                                              const int32_t* seg_lo, const int32_t* seg_rows, int32_t* sorted_rows, int32_t* src,
                                              int32_t* status, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(X && cols && row_span && rows && seg_field && seg_lo && seg_rows && sorted_rows && src && status, SATRANS_E_BADARG,
                    "embed_rows_sort_fields: null pointer");
    SATRANS_REQUIRE(id_dtype == SATRANS_ID_F32 || id_dtype == SATRANS_ID_I32 || id_dtype == SATRANS_ID_I64, SATRANS_E_BADARG,
                    "embed_rows_sort_fields: id_dtype %d", id_dtype);
    SATRANS_REQUIRE(B > 0 && B <= 8192 && F > 0 && F <= kSortFieldsMax, SATRANS_E_UNSUPPORTED,
                    "embed_rows_sort_fields: B=%d F=%d (B <= 8192, F <= %d)", B, F, kSortFieldsMax);
    SortFields sf;
    for (int k = 0; k < F; ++k) {
        SATRANS_REQUIRE(seg_field[k] >= 0 && seg_field[k] < F && seg_rows[k] > 0 && (k == 0 || seg_lo[k] >= seg_lo[k - 1] + seg_rows[k - 1]),
                        SATRANS_E_BADARG, "embed_rows_sort_fields: segment %d is not a table of its own behind segment %d", k, k - 1);
        sf.field[k] = seg_field[k];
        sf.lo[k] = seg_lo[k];
        sf.bits[k] = bits_for(seg_rows[k]);
    }
    for (int k = F; k < kSortFieldsMax; ++k) { sf.field[k] = 0; sf.lo[k] = 0; sf.bits[k] = 1; }
    const SortIds ids{X, id_dtype, x_stride, cols, row_span, rows, status};
    if (B <= 1024) sort_fields_kernel<1, true><<<F, 1024, 0, stream>>>(nullptr, B, F, sf, sorted_rows, src, ids);
    else if (B <= 2048) sort_fields_kernel<2, true><<<F, 1024, 0, stream>>>(nullptr, B, F, sf, sorted_rows, src, ids);
    else if (B <= 4096) sort_fields_kernel<4, true><<<F, 1024, 0, stream>>>(nullptr, B, F, sf, sorted_rows, src, ids);
    else sort_fields_kernel<8, true><<<F, 1024, 0, stream>>>(nullptr, B, F, sf, sorted_rows, src, ids);
    SATRANS_CHECK_LAUNCH("sort_fields_kernel(ids)");
    return SATRANS_OK;
}

extern "C" int64_t satrans_embed_reg_partials(int64_t total_rows, int64_t n, int D) {
    (void)total_rows;
    // run_touched writes three groups of partials behind the streaming kernel's slots: one per apply block, one per chunk
    // block (finish kernel) and one per superchunk block (spans kernel, which writes its slot unconditionally)
    const int64_t supers = ceil_div(ceil_div(n, kChunk), kSuper);
    return kStreamBlocks + touched_blocks(n, D) + chunk_blocks(n, D) + ceil_div(supers * (D / 4), 256);
}

// partial_ws: [chunks][2][D] floats, [chunks] int32 info, [chunks] int32 trail_row, [n][D] run sums, [n] int32 head flags,
// then the same three record arrays for the superchunks, then [chunks][D] + [chunks] int32 for the runs they finish
extern "C" int64_t satrans_embed_partial_ws_floats(int64_t n, int D) {
    const int64_t chunks = ceil_div(n, kChunk), supers = ceil_div(chunks, kSuper);
    return chunks * 2 * D + 2 * chunks + 4 + n * D + n + 4 + supers * 2 * D + 2 * supers + 4 + chunks * D + chunks;
}

// shared by the two entry points below: segmented sums in position order, then either the optimizer step (G == nullptr)
// or a store into the dense gradient buffer G
static int run_touched(float* arena, float* m, float* v, int D, const int32_t* sorted_rows, const int32_t* src, int64_t n,
                       const float* gemb, float* partial_ws, const AdamK& k, double* reg_partials, float* G, int32_t* last,
                       int t, hipStream_t stream) {
    const int64_t chunks = ceil_div(n, kChunk);
    float4* partial = (float4*)partial_ws;
    int32_t* info = (int32_t*)(partial_ws + chunks * 2 * D);
    int32_t* trail_row = info + chunks;
    const int64_t blocks = touched_blocks(n, D), cblocks = chunk_blocks(n, D);
    double* reg_a = reg_partials + kStreamBlocks;
    double* reg_b = reg_a + blocks;
    int64_t off = chunks * 2 * D + 2 * chunks;
    off = (off + 3) & ~(int64_t)3;                     // keep the run sums 16-byte aligned
    float4* rowsum = (float4*)(partial_ws + off);
    int32_t* head_of = (int32_t*)(partial_ws + off + n * D);
    const int64_t supers = ceil_div(chunks, kSuper);
    int64_t off2 = off + n * D + n;
    off2 = (off2 + 3) & ~(int64_t)3;
    float4* partial2 = (float4*)(partial_ws + off2);
    int32_t* info2 = (int32_t*)(partial_ws + off2 + supers * 2 * D);
    int32_t* trail_row2 = info2 + supers;
    const int64_t sblocks = ceil_div(supers * (D / 4), 256);
    int64_t off3 = off2 + supers * 2 * D + 2 * supers;
    off3 = (off3 + 3) & ~(int64_t)3;
    float4* done_sum = (float4*)(partial_ws + off3);
    int32_t* done_row = (int32_t*)(partial_ws + off3 + chunks * D);
    double* reg_c = reg_b + cblocks;
    DISPATCH_LPR(D, (touched_chunks_kernel<LPR><<<(unsigned)cblocks, 256, 0, stream>>>(
                        (float4*)arena, (float4*)m, (float4*)v, sorted_rows, src, n, (const float4*)gemb, partial, info,
                        trail_row, rowsum, head_of, partial2, info2, trail_row2, done_sum, done_row)));
    SATRANS_CHECK_LAUNCH("touched_chunks_kernel");
    if ((256 / (D / 4)) % kSuper != 0) {      // (D = 128: a workgroup holds half a superchunk)
        DISPATCH_LPR(D, (touched_super_kernel<LPR><<<(unsigned)sblocks, 256, 0, stream>>>(
                            chunks, (const float4*)partial, info, trail_row, partial2, info2, trail_row2, done_sum, done_row)));
        SATRANS_CHECK_LAUNCH("touched_super_kernel");
    }
    DISPATCH_LPR(D, (touched_step_kernel<LPR><<<(unsigned)(sblocks + cblocks + blocks), 256, 0, stream>>>(
                        sblocks, cblocks, (float4*)arena, (float4*)m, (float4*)v, k, (float4*)G, last, t, sorted_rows, n,
                        (const float4*)rowsum, head_of, reg_a, chunks, (const float4*)done_sum, done_row, reg_b, supers,
                        (const float4*)partial2, info2, trail_row2, reg_c)));
    SATRANS_CHECK_LAUNCH("touched_step_kernel");
    return SATRANS_OK;
}

extern "C" int satrans_embed_adam_touched(float* arena, float* m, float* v, int D, const int32_t* sorted_rows,
                                          const int32_t* src, int64_t n, const float* gemb, float* partial_ws,
                                          const satrans_adam_hparams* h, double* reg_partials, int32_t* last, int t,
                                          void* stream_) {
    SATRANS_REQUIRE(arena && m && v && sorted_rows && src && gemb && partial_ws && h && reg_partials, SATRANS_E_BADARG,
                    "embed_adam_touched: null pointer");
    SATRANS_REQUIRE(n > 0, SATRANS_E_BADARG, "embed_adam_touched: n=%lld", (long long)n);
    return run_touched(arena, m, v, D, sorted_rows, src, n, gemb, partial_ws, make_adamk(*h), reg_partials, nullptr, last, t,
                       (hipStream_t)stream_);
}

extern "C" int satrans_embed_segment_sums(const int32_t* sorted_rows, const int32_t* src, int64_t n, const float* gemb,
                                          int D, float* partial_ws, double* reg_partials, float* g_rows, void* stream_) {
    SATRANS_REQUIRE(sorted_rows && src && gemb && partial_ws && reg_partials && g_rows, SATRANS_E_BADARG,
                    "embed_segment_sums: null pointer");
    SATRANS_REQUIRE(n > 0, SATRANS_E_BADARG, "embed_segment_sums: n=%lld", (long long)n);
    AdamK k = {};
    return run_touched(nullptr, nullptr, nullptr, D, sorted_rows, src, n, gemb, partial_ws, k, reg_partials, g_rows, nullptr,
                       0, (hipStream_t)stream_);
}

extern "C" int64_t satrans_embed_adam_rows_partials(int64_t rows, int D) { return ceil_div(rows * (D / 4), 256); }

extern "C" int satrans_embed_adam_rows(float* arena, float* m, float* v, int32_t* last, int64_t row0, int64_t rows, int D,
                                       const float* g_rows, const satrans_adam_hparams* h, int t, double* reg_partials,
                                       void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(arena && m && v && g_rows && h && reg_partials, SATRANS_E_BADARG, "embed_adam_rows: null pointer");
    SATRANS_REQUIRE(row0 >= 0 && rows > 0, SATRANS_E_BADARG, "embed_adam_rows: rows=%lld", (long long)rows);
    const AdamK k = make_adamk(*h);
    DISPATCH_LPR(D, (adam_rows_kernel<LPR><<<(unsigned)ceil_div(rows * LPR, 256), 256, 0, stream>>>(
                        (float4*)arena, (float4*)m, (float4*)v, row0, rows, (const float4*)g_rows, last, t, k, reg_partials)));
    SATRANS_CHECK_LAUNCH("adam_rows_kernel");
    return SATRANS_OK;
}

extern "C" int satrans_embed_pack_rows(const int32_t* src, int64_t n, const float* gemb, int D, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(src && gemb && out && n > 0, SATRANS_E_BADARG, "embed_pack_rows: bad arguments");
    DISPATCH_LPR(D, (pack_rows_kernel<LPR><<<(unsigned)ceil_div(n * LPR, 256), 256, 0, stream>>>(src, n, (const float4*)gemb,
                                                                                              (float4*)out)));
    SATRANS_CHECK_LAUNCH("pack_rows_kernel");
    return SATRANS_OK;
}

// touched-row bitmap of an already sorted id list (the every-step streaming Adam skips these rows)
extern "C" int satrans_embed_mark_touched(const int32_t* sorted_rows, int64_t n, int64_t total_rows, uint32_t* touched,
                                          void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(touched && total_rows > 0 && (n == 0 || sorted_rows), SATRANS_E_BADARG, "embed_mark_touched: bad arguments");
    hipError_t e = hipMemsetAsync(touched, 0, sizeof(uint32_t) * (size_t)ceil_div(total_rows, 32), stream);
    SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "embed_mark_touched: memset: %s", hipGetErrorString(e));
    if (n > 0) {
        mark_heads_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(sorted_rows, n, touched);
        SATRANS_CHECK_LAUNCH("mark_heads_kernel");
    }
    return SATRANS_OK;
}

extern "C" int satrans_embed_adam_untouched(float* arena, float* m, float* v, int64_t first_row, int64_t total_rows, int D,
                                            const uint32_t* touched, const satrans_adam_hparams* h, double* reg_partials,
                                            int grid_blocks, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(arena && m && v && touched && h && reg_partials, SATRANS_E_BADARG, "embed_adam_untouched: null pointer");
    SATRANS_REQUIRE(total_rows > 0 && first_row >= 0 && first_row <= total_rows, SATRANS_E_BADARG,
                    "embed_adam_untouched: rows [%lld, %lld)", (long long)first_row, (long long)total_rows);
    const AdamK k = make_adamk(*h);
    const int64_t n4 = total_rows * (D / 4), first4 = first_row * (D / 4);
    // Grid: measured on MI355X (tools/adam_sweep.sh, 841 MB x 3 arrays): 512-768 persistent blocks of 256 threads reach
    // 5.3-5.4 TB/s, 2048 blocks 4.7 TB/s, 256 blocks 3.8 TB/s; the contiguous-slice-per-block form is 6 % slower than the
    // grid-wide stride.  512 also leaves wave slots to the layer kernels when the call runs on a side stream.
    // grid_blocks overrides; partial-sum slots beyond the grid are cleared.
    const int blocks = grid_blocks > 0 ? std::min(grid_blocks, kStreamBlocks) : 512;
    if (blocks < kStreamBlocks) {
        hipError_t e = hipMemsetAsync(reg_partials + blocks, 0, sizeof(double) * (kStreamBlocks - blocks), stream);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "embed_adam_untouched: memset: %s", hipGetErrorString(e));
    }
    // four 16-byte elements in flight per lane, non-temporal accesses (tools/adam_sweep.sh, round 1: 8 in flight, cached
    // accesses and the contiguous-slice form all measured slower)
    DISPATCH_LPR(D, (adam_untouched_kernel<LPR, 4, true><<<blocks, kStreamBlock, 0, stream>>>(
                        (float4*)arena, (float4*)m, (float4*)v, first4, n4, touched, k, reg_partials)));
    SATRANS_CHECK_LAUNCH("adam_untouched_kernel");
    return SATRANS_OK;
}

extern "C" int satrans_embed_grad_dense(const float* arena, const int32_t* sorted_rows, const int32_t* src, int64_t n,
                                        const float* gemb, int64_t total_rows, int D, float l2, float* g_arena,
                                        void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(arena && sorted_rows && src && gemb && g_arena, SATRANS_E_BADARG, "embed_grad_dense: null pointer");
    const int lpr = D / 4;
    DISPATCH_LPR(D, (grad_dense_kernel<LPR><<<(unsigned)ceil_div(n * lpr, 256), 256, 0, stream>>>(
                        sorted_rows, src, n, (const float4*)gemb, (float4*)g_arena)));
    SATRANS_CHECK_LAUNCH("grad_dense_kernel");
    if (l2 != 0.f) {
        const int64_t ne = total_rows * D;
        add_l2_grad_kernel<<<(unsigned)ceil_div(ne, 256), 256, 0, stream>>>(arena, g_arena, ne, 2.0f * l2);
        SATRANS_CHECK_LAUNCH("add_l2_grad_kernel");
    }
    return SATRANS_OK;
}

extern "C" int satrans_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, const satrans_adam_hparams* h,
                                 void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(p && g && m && v && h, SATRANS_E_BADARG, "adam_flat: null pointer");
    if (n <= 0) return SATRANS_OK;
    adam_flat_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(p, g, m, v, n, make_adamk(*h));
    SATRANS_CHECK_LAUNCH("adam_flat_kernel");
    return SATRANS_OK;
}

// satrans_adam_flat followed by out[0] += sum(vals[0..count)), in one launch
extern "C" int satrans_adam_flat_sum(float* p, const float* g, float* m, float* v, int64_t n, const satrans_adam_hparams* h,
                                     const double* vals, int64_t count, double* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(p && g && m && v && h && vals && out && n >= 0 && count >= 0, SATRANS_E_BADARG, "adam_flat_sum: bad arguments");
    adam_flat_sum_kernel<<<(unsigned)ceil_div(n, 1024) + 1, 1024, 0, stream>>>(p, g, m, v, n, make_adamk(*h), vals, count, out);
    SATRANS_CHECK_LAUNCH("adam_flat_sum_kernel");
    return SATRANS_OK;
}

extern "C" int satrans_sum_f64(const double* vals, int64_t count, double* out, int accumulate, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(vals && out && count >= 0, SATRANS_E_BADARG, "sum_f64: bad arguments");
    sum_f64_kernel<<<1, 1024, 0, stream>>>(vals, count, out, accumulate);
    SATRANS_CHECK_LAUNCH("sum_f64_kernel");
    return SATRANS_OK;
}

// ---- lazy-exact path ---------------------------------------------------------------------------------------------
constexpr int kFlushBlocks = 4096;

extern "C" int64_t satrans_embed_lazy_reg_partials(int64_t n, int D) { return ceil_div(n * D, 256) + kFlushBlocks; }

#define DISPATCH_D(D, CALL)                                                          \
    switch (D) {                                                                     \
        case 16: { constexpr int DD = 16; CALL; } break;                             \
        case 32: { constexpr int DD = 32; CALL; } break;                             \
        case 64: { constexpr int DD = 64; CALL; } break;                             \
        case 128: { constexpr int DD = 128; CALL; } break;                           \
        default:                                                                     \
            ::satrans::set_error("embedding_dim %d not in {16,32,64,128}", (D));     \
            return SATRANS_E_UNSUPPORTED;                                            \
    }

// Replays the pending regulariser-only steps (last[r], target] of every distinct row in sorted_rows.
// table [>= target + 1] double2: table[s] = (fp32(lr / (1 - beta1^s)), 1 / (double)fp32(sqrt(1 - beta2^s))); h supplies beta1, beta2, eps, l2.
// reg_partials: satrans_embed_lazy_reg_partials(n, D) doubles; this call writes the first ceil(n*D/256).
extern "C" int satrans_embed_lazy_replay(float* arena, float* m, float* v, int32_t* last, int D,
                                         const int32_t* sorted_rows, int64_t n, int target, const double* table,
                                         const satrans_adam_hparams* h, double* reg_partials, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(arena && m && v && last && sorted_rows && table && h && reg_partials, SATRANS_E_BADARG,
                    "embed_lazy_replay: null pointer");
    SATRANS_REQUIRE(n > 0 && target >= 0, SATRANS_E_BADARG, "embed_lazy_replay: n=%lld target=%d", (long long)n, target);
    const LazyK k = make_lazyk(*h);
    SATRANS_REQUIRE(D % 4 == 0, SATRANS_E_UNSUPPORTED, "embed_lazy_replay: embedding_dim %d is not a multiple of 4", D);
    const int64_t slots = ceil_div(n * D, 256);
    const int64_t blocks = ceil_div(n * (D / 4), 256);
    DISPATCH_D(D, (lazy_replay_kernel<DD><<<(unsigned)blocks, 256, 0, stream>>>((float4*)arena, (float4*)m, (float4*)v, last,
                                                                                sorted_rows, n, target, (const double2*)table, k,
                                                                                reg_partials, slots)));
    SATRANS_CHECK_LAUNCH("lazy_replay_kernel");
    return SATRANS_OK;
}

// Brings EVERY row to step `target`.  Writes reg_partials[ceil(n*D/256) ...+kFlushBlocks) where n is the batch row count
// the workspace was sized for (pass the same n).
extern "C" int satrans_embed_lazy_flush(float* arena, float* m, float* v, int32_t* last, int64_t total_rows, int D,
                                        int target, const double* table, const satrans_adam_hparams* h, int64_t n,
                                        double* reg_partials, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(arena && m && v && last && table && h && reg_partials, SATRANS_E_BADARG, "embed_lazy_flush: null pointer");
    const LazyK k = make_lazyk(*h);
    double* reg = reg_partials + ceil_div(n * D, 256);
    DISPATCH_D(D, (lazy_flush_kernel<DD><<<kFlushBlocks, 256, 0, stream>>>((float4*)arena, (float4*)m, (float4*)v, last, total_rows,
                                                                          target, (const double2*)table, k, reg)));
    SATRANS_CHECK_LAUNCH("lazy_flush_kernel");
    return SATRANS_OK;
}

// Diagnostic: mismatches of the replay's packed sqrt (mode 0, bit patterns [first, first + count)) or division (mode 1, `count`
// pairs from seed `first`) against the IEEE operations; *mismatches (host) receives the count.
extern "C" int satrans_debug_check_packed_math(int mode, uint64_t first, uint64_t count, uint64_t* mismatches, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(mode >= 0 && mode <= 3 && mismatches, SATRANS_E_BADARG, "check_packed_math: bad arguments");
    unsigned long long* dev = nullptr;
    SATRANS_REQUIRE(hipMalloc(&dev, sizeof(unsigned long long)) == hipSuccess, SATRANS_E_LAUNCH, "check_packed_math: hipMalloc");
    (void)hipMemsetAsync(dev, 0, sizeof(unsigned long long), stream);
    packed_math_check_kernel<<<4096, 256, 0, stream>>>(mode, first, count, dev);
    unsigned long long host = 0;
    const hipError_t e = hipMemcpyAsync(&host, dev, sizeof(host), hipMemcpyDeviceToHost, stream);
    (void)hipStreamSynchronize(stream);
    (void)hipFree(dev);
    SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "check_packed_math: %s", hipGetErrorString(e));
    *mismatches = host;
    return SATRANS_OK;
}

// last[r] = t for every distinct row of sorted_rows (after the touched-row Adam of step t)
extern "C" int satrans_embed_lazy_mark(const int32_t* sorted_rows, int64_t n, int32_t* last, int t, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(sorted_rows && last && n > 0, SATRANS_E_BADARG, "embed_lazy_mark: bad arguments");
    mark_last_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(sorted_rows, n, last, t);
    SATRANS_CHECK_LAUNCH("mark_last_kernel");
    return SATRANS_OK;
}


// ---- the other optimizers of compile(): dense elementwise steps (see include/satrans_hip.h) ---------------------------------------
namespace satrans {
__global__ void optim_flat_kernel(int kind, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ st, int64_t n,
                                  float lr, float alpha, float eps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    float pi = p[i];
    if (kind == SATRANS_OPT_SGD) {
        pi = __fadd_rn(pi, __fmul_rn(-lr, gi));                                   // p.add_(g, alpha=-lr)
    } else {
        float s = st[i];
        if (kind == SATRANS_OPT_ADAGRAD) s = fmaf(gi, gi, s);                      // state_sum.addcmul_(g, g, value=1)
        else s = fmaf(__fmul_rn(1.0f - alpha, gi), gi, __fmul_rn(s, alpha));       // square_avg.mul_(alpha).addcmul_(g, g, value=1-alpha)
        st[i] = s;
        const float denom = __fadd_rn(sqrtf(s), eps);                         // std = sqrt(.) + eps
        pi = __fadd_rn(pi, __fmul_rn(-lr, __fdiv_rn(gi, denom)));                 // p.addcdiv_(g, std, value=-lr)
    }
    p[i] = pi;
}
}  // namespace satrans

extern "C" int satrans_optim_flat(int kind, float* p, const float* g, float* state, int64_t n, float lr, float alpha, float eps,
                                  void* stream_) {
    SATRANS_REQUIRE(kind >= SATRANS_OPT_SGD && kind <= SATRANS_OPT_RMSPROP, SATRANS_E_BADARG, "optim_flat: kind %d", kind);
    SATRANS_REQUIRE(p && g && n > 0 && (kind == SATRANS_OPT_SGD || state), SATRANS_E_BADARG, "optim_flat: bad arguments");
    satrans::optim_flat_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, (hipStream_t)stream_>>>(kind, p, g, state, n, lr, alpha, eps);
    SATRANS_CHECK_LAUNCH("optim_flat_kernel");
    return SATRANS_OK;
}
