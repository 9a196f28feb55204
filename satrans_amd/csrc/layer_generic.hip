// Meta_Transformer_Layer forward and backward for shapes the fused kernels are not built for - first of all BASELINE
// configs[4]: 64 fields, embedding_dim 64, MetaNet hidden 128 (generated row 16,384 floats), where the weights alone
// (138 KB fp32) leave no room in LDS for a token tile.  Reference: models/satrans.py:50-100, models/submodules.py:77-103.
//
// Unfused: a layer is a short sequence of launches over token rows kept in HBM in SCENARIO-SORTED sample order (position p
// holds sample order[p]; a scenario's tokens are one contiguous row range, so a product with that scenario's generated
// weights is a grouped GEMM):
//     permute-in -> {q0,k0,v} = x W            (one batched f32-MFMA GEMM)
//                -> per role: h = relu(z W1[s]), m = h W2[s] (grouped GEMMs), q = LN(drop(m) + z)
//                -> attention per (sample, head): MFMA kernel (F <= 64, d = 16) or one lane per query row ("wavefront")
//                -> u = o Wo^T -> y = LN(drop(u) + x), written back in the caller's sample order
// and the backward mirrors it with the forward activations SAVED (288 GB of HBM: 14 token-row buffers per layer) instead of
// recomputed; weight gradients are token-contraction GEMMs (split over row chunks, partials summed in a fixed order: no
// float atomics, bitwise reproducible).  At D = 64 the arithmetic intensity of these K = 64 / 128 products is ~40 MAC per
// float moved, so the path is HBM-bound by construction (~45 token-row buffers read or written per layer and direction); it
// is the general, correct path, the fused kernels remain the fast one for the shapes they cover.
#include "layer_fused_common.h"

namespace satrans {

constexpr int kTnRows = 512;                        // token rows per workgroup of a weight-gradient product (2048: 1.4x slower)

struct GemmBatch {      // blockIdx.z selects one of up to three products that share shapes (q0 / k0 / v)
    const float* A[3];
    const float* B[3];
    float* C[3];
};

// C[m][n] (op)= sum_k A[m][k] B(k, n) over the rows of one segment.  Segment s = rows [seg[s] F, seg[s+1] F) and takes
// B + s * b_seg_stride (seg == nullptr: one segment of M rows).  TRANSB: B(k, n) = B[n * ldb + k], else B[k * ldb + n].
// EPI: 0 store, 1 relu then store, 2 C += acc (C = mask + acc when a mask pointer is given), 3 store where mask[m][n] > 0
// else 0 (relu backward).
// K, N multiples of 16, <= 128; A and C are dense token-row buffers (lda = K, ldc = N).
//
// HBM-bound by shape (K <= 128: ~40 MAC per float moved), so the kernel is built around the row traffic: no LDS for the token
// rows at all.  The product is computed TRANSPOSED, C^T = B^T A^T: the MFMA's first operand is a B^T fragment from LDS (the
// weights of the segment, staged once per workgroup), the second is A^T - lane (m, g) supplies A[m][16 j + 4 g + i], which
// is component i of ONE 16-byte load per 16 k - and the accumulator tile then holds C[m][16 jn + 4 g + r], four consecutive
// columns of one row per lane: 16-byte loads and 16-byte stores straight between HBM and the MFMA operands, the next row
// group's loads in flight under the current group's MFMAs.  The k order of the contraction is permuted the same way on both
// operands (k = 16 j + 4 g + i), which a sum does not see.
// B^T in LDS as [g][n][j][i] planes (plane stride a multiple of 256 B, row stride an ODD number of 16-byte slots): the
// 16 lanes of a ds_read_b128 service group hold 16 different n, i.e. 16 different slots - conflict-free.
constexpr int kG2Rows = 512;     // token rows per workgroup (4 waves x 8 groups of 16)

__host__ __device__ inline int g2_row_slots(int K) { return (K >> 4) | 1; }
__host__ __device__ inline int g2_plane_floats(int K, int N) { return ((N * g2_row_slots(K) * 4 + 63) / 64) * 64; }

// Stage B^T as planes: element (k, nn) -> plane (k >> 2) & 3, row nn, position 4 (k >> 4) + (k & 3).  16-byte global loads; when k
// is the contiguous index of the source (KCONTIG: B(k, nn) = B[nn * ldb + k]) the four values are one 16-byte LDS write too.
template <bool KCONTIG>
__device__ __forceinline__ void g2_stage(const float* __restrict__ B, int K, int N, int ldb, float* __restrict__ img, int RS, int PL) {
    if (KCONTIG) {
        const int kq = K >> 2;
        for (int e = threadIdx.x; e < N * kq; e += blockDim.x) {
            const int nn = e / kq, k = (e - nn * kq) * 4;
            const float4 v = *reinterpret_cast<const float4*>(B + (size_t)nn * ldb + k);
            *reinterpret_cast<float4*>(img + ((k >> 2) & 3) * PL + nn * RS + 4 * (k >> 4)) = v;
        }
    } else {
        const int nq = N >> 2;
        for (int e = threadIdx.x; e < K * nq; e += blockDim.x) {
            const int k = e / nq, nn = (e - k * nq) * 4;
            const float4 v = *reinterpret_cast<const float4*>(B + (size_t)k * ldb + nn);
            float* dst = img + ((k >> 2) & 3) * PL + nn * RS + 4 * (k >> 4) + (k & 3);
            dst[0] = v.x; dst[RS] = v.y; dst[2 * RS] = v.z; dst[3 * RS] = v.w;
        }
    }
}

template <bool TRANSB, int EPI, int KJ>       // KJ = K / 16
__global__ __launch_bounds__(256) void gen_gemm_kernel(GemmBatch gb, const int32_t* __restrict__ seg, int M, int F, int N,
                                                     int ldb, int64_t b_seg_stride, const float* __restrict__ mask) {
    extern __shared__ __align__(16) float g2_lds[];
    constexpr int K = KJ * 16;
    const int s = blockIdx.y;
    const int64_t r_lo = seg ? (int64_t)seg[s] * F : 0, r_hi = seg ? (int64_t)seg[s + 1] * F : M;
    const int64_t row0 = r_lo + (int64_t)blockIdx.x * kG2Rows;
    if (row0 >= r_hi) return;
    const float* __restrict__ A = gb.A[blockIdx.z];
    const float* __restrict__ B = gb.B[blockIdx.z] + (size_t)s * b_seg_stride;
    float* __restrict__ C = gb.C[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int RS = g2_row_slots(K) * 4, PL = g2_plane_floats(K, N);
    g2_stage<TRANSB>(B, K, N, ldb, g2_lds, RS, PL);
    __syncthreads();
    const float* bt = g2_lds + g * PL + n * RS;
    const int NT = N >> 4;
    float4 a[KJ], an[KJ];
    auto load_rows = [&](int64_t g0, float4 (&dst)[KJ]) {
        const int64_t row = g0 + n;
        if (row < r_hi) {
            const float4* src = reinterpret_cast<const float4*>(A + row * K + 4 * g);
#pragma unroll
            for (int j = 0; j < KJ; ++j) dst[j] = src[4 * j];
        } else {
#pragma unroll
            for (int j = 0; j < KJ; ++j) dst[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int64_t wave_row0 = row0 + 16 * wave;
    const int64_t tile_hi = min(r_hi, row0 + kG2Rows);
    if (wave_row0 < tile_hi) load_rows(wave_row0, a);
    for (int64_t g0 = wave_row0; g0 < tile_hi; g0 += 64) {
        const bool more = g0 + 64 < tile_hi;
        if (more) load_rows(g0 + 64, an);
        const int64_t row = g0 + n;
        for (int jn = 0; jn < NT; ++jn) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* bp = bt + 16 * jn * RS;
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const float4 b = *reinterpret_cast<const float4*>(bp + 4 * j);
                acc = mfma4(b.x, a[j].x, acc);
                acc = mfma4(b.y, a[j].y, acc);
                acc = mfma4(b.z, a[j].z, acc);
                acc = mfma4(b.w, a[j].w, acc);
            }
            if (row < r_hi) {
                float4* dst = reinterpret_cast<float4*>(C + row * N + 16 * jn + 4 * g);
                float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
                if (EPI == 1) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                if (EPI == 2) {      // the addend is C itself, or `mask` when given (C = mask + acc: saves a copy of the rows)
                    const float4 o = mask ? *reinterpret_cast<const float4*>(mask + row * N + 16 * jn + 4 * g) : *dst;
                    v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
                }
                if (EPI == 3) {
                    const float4 mk = *reinterpret_cast<const float4*>(mask + row * N + 16 * jn + 4 * g);
                    v = make_float4(mk.x > 0.f ? v.x : 0.f, mk.y > 0.f ? v.y : 0.f, mk.z > 0.f ? v.z : 0.f, mk.w > 0.f ? v.w : 0.f);
                }
                *dst = v;
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < KJ; ++j) a[j] = an[j];
        }
    }
}

// C[m][n] = (addend ? addend[m][n] : 0) + sum over the NSRC sources i of sum_k A_i[m][k] W_i[n][k]   (gradient of the rows through
// NSRC projections that read the same input: dx = dr + gq Wq^T + gk Wk^T + gv Wv^T in ONE pass - the sources are read once,
// the sum is written once, where three accumulating products read and write the sum three times).  Same operand scheme as
// gen_gemm_kernel (transposed product, 16-byte loads and stores, B^T planes in LDS, next row group's loads in flight); K = N = D.
template <int KJ, int NSRC>
__global__ __launch_bounds__(256) void gen_gemm_sum_kernel(GemmBatch gb, int nsrc_unused, int M, const float* __restrict__ addend,
                                                         float* __restrict__ C) {
    extern __shared__ __align__(16) float g2_lds[];
    constexpr int K = KJ * 16, N = K;
    const int64_t row0 = (int64_t)blockIdx.x * kG2Rows;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int RS = g2_row_slots(K) * 4, PL = g2_plane_floats(K, N);
#pragma unroll
    for (int i = 0; i < NSRC; ++i) {
        g2_stage<true>(gb.B[i], K, N, K, g2_lds + (size_t)i * 4 * PL, RS, PL);           // B(k, nn) = W[nn][k]
    }
    __syncthreads();
    float4 a[NSRC][KJ], an[NSRC][KJ];
    auto load_rows = [&](int64_t g0, float4 (&dst)[NSRC][KJ]) {
        const int64_t row = g0 + n;
#pragma unroll
        for (int i = 0; i < NSRC; ++i) {
            if (row < M) {
                const float4* src = reinterpret_cast<const float4*>(gb.A[i] + row * K + 4 * g);
#pragma unroll
                for (int j = 0; j < KJ; ++j) dst[i][j] = src[4 * j];
            } else {
#pragma unroll
                for (int j = 0; j < KJ; ++j) dst[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    const int64_t wave_row0 = row0 + 16 * wave;
    const int64_t tile_hi = min((int64_t)M, row0 + kG2Rows);
    if (wave_row0 < tile_hi) load_rows(wave_row0, a);
    for (int64_t g0 = wave_row0; g0 < tile_hi; g0 += 64) {
        const bool more = g0 + 64 < tile_hi;
        if (more) load_rows(g0 + 64, an);
        const int64_t row = g0 + n;
        for (int jn = 0; jn < KJ; ++jn) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NSRC; ++i) {
                const float* bp = g2_lds + (size_t)i * 4 * PL + g * PL + (16 * jn + n) * RS;
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    const float4 b = *reinterpret_cast<const float4*>(bp + 4 * j);
                    acc = mfma4(b.x, a[i][j].x, acc);
                    acc = mfma4(b.y, a[i][j].y, acc);
                    acc = mfma4(b.z, a[i][j].z, acc);
                    acc = mfma4(b.w, a[i][j].w, acc);
                }
            }
            if (row < M) {
                float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
                if (addend) {
                    const float4 o = *reinterpret_cast<const float4*>(addend + row * N + 16 * jn + 4 * g);
                    v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
                }
                *reinterpret_cast<float4*>(C + row * N + 16 * jn + 4 * g) = v;
            }
#ifndef SATRANS_EXP_SUM_NOSB
            // (the weight fragments of later tiles stay behind this tile: hoisted, all 12 KJ reads of a row group held 320 registers
            //  - one wave per SIMD; with the barrier 2-3 waves, profiles/r06_c5_fusions.txt)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < NSRC; ++i)
#pragma unroll
                for (int j = 0; j < KJ; ++j) a[i][j] = an[i][j];
        }
    }
}

// partial[(s * chunks + c)][k][n] = sum over the token rows of chunk c of segment s of A[m][k] G[m][n]   (K, N multiples of 16,
// KA = ceil(K / 64) and NA = ceil(N / 64) 64-column blocks, KA * NA <= 2).  Chunks beyond a segment's rows write nothing.
//
// Token rows straight from HBM into the MFMA operands: lane (c, g) loads the 16 bytes A[m = 4 step + g][4 c .. 4 c + 3] (the 16
// lanes of a g cover one 256-byte row: fully coalesced), and component i of that load is the first operand of the output
// tile whose 16 rows are k = 4 c' + i (c' = 0..15) - a row PERMUTATION of dW that the store undoes; the same for G and the
// columns.  16 KA NA MFMAs per two or three 16-byte loads, no LDS on the operand path, the next step's loads issued before
// the current step's MFMAs.  The four waves of a workgroup own a quarter of the chunk's rows each and add their tiles into
// one LDS image in wave order (fixed order => bitwise reproducible), which is written out coalesced.
// FAN: the NA column blocks are NA separate gradient tensors G, G1, G2 of width N <= 64 that share A (dWq, dWk, dWv = x^T gq, x^T gk,
// x^T gv: x is read once); the partial of a chunk is then [NA][K][N].
template <int KA, int NA, bool FAN = false>
__global__ __launch_bounds__(256, KA * NA >= 3 ? 1 : 2) void gen_gemm_tn_kernel(const float* __restrict__ A, const float* __restrict__ G,
                                                        const int32_t* __restrict__ seg, int M, int F, int K, int N, int lda,
                                                        float* __restrict__ partial, const float* __restrict__ G1 = nullptr,
                                                        const float* __restrict__ G2 = nullptr) {
    extern __shared__ __align__(16) float tn_lds[];      // [K][N]  (FAN: [NA][K][N])
    const int s = blockIdx.y, cidx = blockIdx.x, chunks = gridDim.x;
    const int64_t r_lo = seg ? (int64_t)seg[s] * F : 0, r_hi = seg ? (int64_t)seg[s + 1] * F : M;
    const int64_t c_lo = r_lo + (int64_t)cidx * kTnRows, c_hi = min(r_hi, c_lo + kTnRows);
    if (c_lo >= r_hi) return;          // the reducer reads only the chunks a segment uses
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    f32x4 acc[4 * KA][4 * NA];
#pragma unroll
    for (int i = 0; i < 4 * KA; ++i)
#pragma unroll
        for (int j = 0; j < 4 * NA; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int kWaveRows = kTnRows / 4;
    const int64_t w_lo = c_lo + (int64_t)wave * kWaveRows, w_hi = min(c_hi, w_lo + kWaveRows);
    bool ka_ok[KA], na_ok[NA];
#pragma unroll
    for (int u = 0; u < KA; ++u) ka_ok[u] = 64 * u + 4 * c < K;
#pragma unroll
    for (int u = 0; u < NA; ++u) na_ok[u] = (FAN ? 0 : 64 * u) + 4 * c < N;
    const float* Gp[3] = {G, G1, G2};
    auto load = [&](int64_t m0, float4 (&av)[KA], float4 (&gv)[NA]) {
        const int64_t m = m0 + g;
        const bool in = m < w_hi;
#pragma unroll
        for (int u = 0; u < KA; ++u)
            av[u] = in && ka_ok[u] ? *reinterpret_cast<const float4*>(A + m * lda + 64 * u + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < NA; ++u)
            gv[u] = in && na_ok[u] ? *reinterpret_cast<const float4*>((FAN ? Gp[u < 3 ? u : 0] : G) + m * N + (FAN ? 0 : 64 * u) + 4 * c)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // two steps of loads in flight under the MFMAs of the current one (a step is 16 KA NA MFMAs = 0.2 - 0.4 us, the HBM latency
    // under load is above that)
    float4 av[3][KA], gv[3][NA];
    load(w_lo, av[0], gv[0]);
    load(w_lo + 4, av[1], gv[1]);
    for (int64_t m0 = w_lo; m0 < w_hi; m0 += 12) {
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            if (m0 + 4 * ph < w_hi) {           // wave-uniform
                load(m0 + 4 * ph + 8, av[(ph + 2) % 3], gv[(ph + 2) % 3]);
#pragma unroll
                for (int u = 0; u < KA; ++u) {
                    const float ac[4] = {av[ph][u].x, av[ph][u].y, av[ph][u].z, av[ph][u].w};
#pragma unroll
                    for (int w = 0; w < NA; ++w) {
                        const float gc[4] = {gv[ph][w].x, gv[ph][w].y, gv[ph][w].z, gv[ph][w].w};
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[4 * u + i][4 * w + j] = mfma4(ac[i], gc[j], acc[4 * u + i][4 * w + j]);
                    }
                }
            }
        }
    }
    // tile (u, i) x (w, j): accumulator register r of lane (c, g) is dW[k = 64 u + 4 (4 g + r) + i][n = 64 w + 4 c + j]
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int u = 0; u < KA; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 64 * u + 4 * (4 * g + r) + i;
#pragma unroll
                        for (int w = 0; w < NA; ++w) {
                            const int nn = (FAN ? 0 : 64 * w) + 4 * c;
                            if (k < K && nn < N) {
                                float4* dst = reinterpret_cast<float4*>(tn_lds + (FAN ? w * K * N : 0) + k * N + nn);
                                float4 v = make_float4(acc[4 * u + i][4 * w][r], acc[4 * u + i][4 * w + 1][r],
                                                       acc[4 * u + i][4 * w + 2][r], acc[4 * u + i][4 * w + 3][r]);
                                if (turn > 0) {
                                    const float4 o = *dst;
                                    v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
                                }
                                *dst = v;
                            }
                        }
                    }
        }
        __syncthreads();
    }
    const int psz = (FAN ? NA : 1) * K * N;
    float4* out = reinterpret_cast<float4*>(partial + ((size_t)s * chunks + cidx) * psz);
    for (int e = tid; e < psz >> 2; e += 256) out[e] = reinterpret_cast<const float4*>(tn_lds)[e];
}

// dst[s * dst_seg_stride + e] += sum_c partial[(s * chunks + c)][e] over the chunks segment s uses; S segments (grid.y).  Blocks of 32 elements x 32 groups:
// every group adds its contiguous share of the chunks in chunk order, the group sums are combined in group order (fixed
// order => bitwise reproducible)
__global__ __launch_bounds__(1024) void gen_tn_reduce_kernel(const float* __restrict__ partial, int chunks, int count,
                                                          float* __restrict__ dst, int64_t dst_seg_stride,
                                                          const int32_t* __restrict__ seg, int F, int chunk_rows,
                                                          int chunk_stride = 0) {      // floats between chunks (0: count)
    constexpr int G = 32;                 // groups: 1024 threads (8 groups: four times as many dependent rounds of loads)
    __shared__ float s_part[G][32];
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + lane;
    const int s = blockIdx.y;
    // a segment fills only its first ceil(rows / chunk_rows) partials; the rest were never written
    const int used = seg ? (int)(((int64_t)(seg[s + 1] - seg[s]) * F + chunk_rows - 1) / chunk_rows) : chunks;
    const int share = (used + G - 1) / G;
    const int c0 = grp * share, c1 = min(used, c0 + share);
    float t = 0.f;
    if (e < count) {
        const int stride = chunk_stride ? chunk_stride : count;
        const float* p = partial + (size_t)s * chunks * stride + e;
#pragma unroll 4
        for (int c = c0; c < c1; ++c) t += p[(size_t)c * stride];
    }
    s_part[grp][lane] = t;
    __syncthreads();
    if (grp == 0 && e < count) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < G; ++k) r += s_part[k][lane];
        dst[(size_t)s * dst_seg_stride + e] += r;
    }
}

// dst[e] += sum over the USED slots of part[(s * gridx + c)][e] (segment s uses its first ceil(rows_s / chunk_rows) slots), count <= 256
// values per slot: 32 columns x 32 groups per block, group g takes the used slots with (running index) % 32 == g in order, the
// group sums are combined in group order - fixed order, and the empty slots of the (row tile, segment) grid are never read.
__global__ __launch_bounds__(1024) void gen_slot_reduce_kernel(const float* __restrict__ part, int gridx, int S,
                                                            const int32_t* __restrict__ seg, int F, int chunk_rows, int count,
                                                            float* __restrict__ dst) {
    constexpr int G = 32;                 // groups of 32 columns: 1024 threads
    __shared__ float s_part[G][32];
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + lane;
    float t = 0.f;
    int running = 0;
    for (int s = 0; s < S; ++s) {
        const int used = (int)(((int64_t)(seg[s + 1] - seg[s]) * F + chunk_rows - 1) / chunk_rows);
        if (e < count) {
            const float* p = part + (size_t)s * gridx * count + e;
            for (int c = (grp - running % G + G) % G; c < used; c += G) t += p[(size_t)c * count];
        }
        running += used;
    }
    s_part[grp][lane] = t;
    __syncthreads();
    if (grp == 0 && e < count) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < G; ++k) r += s_part[k][lane];
        dst[e] += r;
    }
}

// ---- all reductions of one layer's backward in ONE launch (round 6) -------------------------------------------------------------
// A layer backward of the general path makes 9 weight-gradient partial sets and 2 gamma / beta slot sets; reducing each behind its
// product was 11 launches of ~9-16 us with nothing else to run (66 per step at configs[4]).  With a partial region of its own
// per product the reductions have no order among each other (every one adds into its own destination), so they are collected
// and run as blockIdx.z of one launch at the end of the layer - the same sums in the same order, bit for bit.
struct ReduceJob {
    const float* partial;
    float* dst;
    const int32_t* seg;
    int64_t dst_seg_stride;
    int chunks, count, F, chunk_rows, chunk_stride, segs, kind;      // kind 0: gen_tn_reduce, 1: gen_slot_reduce (chunks = gridx, segs = S)
};
constexpr int kMaxReduceJobs = 24;
struct ReduceJobs {
    ReduceJob j[kMaxReduceJobs];
};
struct ReduceDefer {          // host side: the jobs collected so far and the partial region they carve their buffers from
    ReduceJobs jobs;
    int n = 0;
    float* part = nullptr;
    int64_t used = 0, cap = 0;
    float* take(int64_t floats) {
        float* r = part + used;
        used += (floats + 3) & ~(int64_t)3;
        return used <= cap ? r : nullptr;
    }
};

__global__ __launch_bounds__(1024) void gen_multi_reduce_kernel(ReduceJobs jobs) {
    constexpr int G = 32;
    __shared__ float s_part[G][32];
    const ReduceJob& jb = jobs.j[blockIdx.z];
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + lane;
    if (blockIdx.x * 32 >= jb.count) return;
    float t = 0.f;
    if (jb.kind == 0) {
        const int s = blockIdx.y;
        if (s >= jb.segs) return;
        const int used = jb.seg ? (int)(((int64_t)(jb.seg[s + 1] - jb.seg[s]) * jb.F + jb.chunk_rows - 1) / jb.chunk_rows) : jb.chunks;
        const int share = (used + G - 1) / G;
        const int c0 = grp * share, c1 = min(used, c0 + share);
        if (e < jb.count) {
            const int stride = jb.chunk_stride ? jb.chunk_stride : jb.count;
            const float* p = jb.partial + (size_t)s * jb.chunks * stride + e;
#pragma unroll 4
            for (int c = c0; c < c1; ++c) t += p[(size_t)c * stride];
        }
        s_part[grp][lane] = t;
        __syncthreads();
        if (grp == 0 && e < jb.count) {
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < G; ++k) r += s_part[k][lane];
            jb.dst[(size_t)s * jb.dst_seg_stride + e] += r;
        }
    } else {
        if (blockIdx.y != 0) return;
        int running = 0;
        for (int s = 0; s < jb.segs; ++s) {
            const int used = (int)(((int64_t)(jb.seg[s + 1] - jb.seg[s]) * jb.F + jb.chunk_rows - 1) / jb.chunk_rows);
            if (e < jb.count) {
                const float* p = jb.partial + (size_t)s * jb.chunks * jb.count + e;
                for (int c = (grp - running % G + G) % G; c < used; c += G) t += p[(size_t)c * jb.count];
            }
            running += used;
        }
        s_part[grp][lane] = t;
        __syncthreads();
        if (grp == 0 && e < jb.count) {
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < G; ++k) r += s_part[k][lane];
            jb.dst[e] += r;
        }
    }
}

// rows in scenario-sorted order <- the layer input in the caller's order (or the arena rows of the fused gather), and back
__global__ void gen_permute_in_kernel(satrans_layer_desc a, const float* __restrict__ src, float* __restrict__ dst, bool is_x) {
    const int q4 = a.D >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)a.B * a.F * q4) return;
    const int64_t m = i / q4;
    const int c = (int)(i - m * q4) * 4;
    const int p = (int)(m / a.F), f = (int)(m - (int64_t)p * a.F);
    const int b = a.order[p];
    const float* row = is_x ? layer_x_row(a, b, f, a.F, a.D) : src + ((size_t)b * a.F + f) * a.D;
    *reinterpret_cast<float4*>(dst + m * a.D + c) = *reinterpret_cast<const float4*>(row + c);
}
__global__ void gen_permute_out_kernel(satrans_layer_desc a, const float* __restrict__ src, float* __restrict__ dst) {
    const int q4 = a.D >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)a.B * a.F * q4) return;
    const int64_t m = i / q4;
    const int c = (int)(i - m * q4) * 4;
    const int p = (int)(m / a.F), f = (int)(m - (int64_t)p * a.F);
    *reinterpret_cast<float4*>(dst + ((size_t)a.order[p] * a.F + f) * a.D + c) = *reinterpret_cast<const float4*>(src + m * a.D + c);
}

// ---- flags `gate` and `bilinear` (satrans.py:61-64,68-69,79-81): the scenario's generated row modulates q0 / k0 directly ----------
// gate:     out[t][c] = z[t][c] * vec_s[c] * 2            (vec = the row, length D)
// bilinear: out[t][h d + e2] = sum_e z[t][h d + e] M_s[h][e][e2]   (M = the row viewed as [H][d][d]; queries only)
// TRANSPOSE = the backward of the same map with respect to z (gate: identical; bilinear: M^T).  Rows are scenario-sorted: the
// scenario of token row t is the segment that holds sample position t / F (S is small: linear scan).
template <bool BILINEAR, bool TRANSPOSE>
__global__ void gen_modulate_kernel(const float* __restrict__ z, const float* __restrict__ tab, int64_t tab_stride,
                                    const int32_t* __restrict__ seg, int S, int F, int D, int dh, float* __restrict__ out, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    const int64_t t = i / D;
    const int c = (int)(i - t * D);
    const int pos = (int)(t / F);
    int s = 0;
    while (s + 1 < S && seg[s + 1] <= pos) ++s;
    const float* row = tab + (size_t)s * tab_stride;
    if (!BILINEAR) {
        out[i] = z[i] * row[c] * 2.0f;
    } else {
        const int h = c / dh, e_ = c - h * dh;
        const float* zr = z + t * D + h * dh;
        float acc = 0.f;
        for (int e = 0; e < dh; ++e)
            acc = fmaf(zr[e], TRANSPOSE ? row[(h * dh + e_) * dh + e] : row[(h * dh + e) * dh + e_], acc);
        out[i] = acc;
    }
}

// gradient of the generated rows from full[s][D][D] = z^T g of the segment (a weight-gradient product): gate takes twice the
// diagonal, bilinear the H diagonal d x d blocks
template <bool BILINEAR>
__global__ void gen_modulate_extract_kernel(const float* __restrict__ full, int S, int D, int dh, float* __restrict__ g_tab,
                                            int64_t tab_stride) {
    const int len = BILINEAR ? D * dh : D;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * len) return;
    const int s = i / len, e_ = i - s * len;
    const float* f = full + (size_t)s * D * D;
    float v;
    if (!BILINEAR) {
        v = 2.0f * f[e_ * D + e_];
    } else {
        const int h = e_ / (dh * dh), rem = e_ - h * dh * dh;
        const int e = rem / dh, e2 = rem - e * dh;
        v = f[(h * dh + e) * D + h * dh + e2];
    }
    g_tab[(size_t)s * tab_stride + e_] += v;
}

struct GenDrop {
    bool on;
    float scale;
    uint32_t thresh, key;
};

// ---- MetaNet forward in one pass (submodules.py:77-103): h = relu(z W1[s]), m = h W2[s], t = drop(m) + z, y = LayerNorm(t) -----------
// Operand scheme of gen_gemm_kernel, chained: the accumulator tile of the first product - lane (token, g) holds h[token][16 ju + 4 g
// .. + 3] - is, as it stands, the second operand of the contraction over u in the second product (the same k permutation
// 16 j + 4 g + i on both sides), and the tiles of m meet the z fragments the lane loaded at the start (the residual) column for
// column; a token's row is spread over the four lanes g, so the LayerNorm statistics are two cross-lane adds.  Read z once,
// write h and t (the backward needs them) and y: 670 MB at the configs[4] shape where the three separate launches move 1.6 GB.
// W1^T and W2^T planes of the scenario in LDS.  norm == false: y = t (MetaNet(use_norm=False)).
template <int KJ, int UJ, int WV>       // D = 16 KJ, U = 16 UJ, WV waves per workgroup (128 WV token rows)
__global__ __launch_bounds__(64 * WV, WV == 8 ? 4 : 2) void gen_metanet_fwd_kernel(const float* __restrict__ z, const float* __restrict__ tab, int64_t tab_stride,
                                                            const int32_t* __restrict__ seg, int M, int F, float* __restrict__ hbuf,
                                                            float* __restrict__ tbuf, float* __restrict__ y,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const int32_t* __restrict__ order, GenDrop dc, bool norm) {
    extern __shared__ __align__(16) float g2_lds[];
    constexpr int D = KJ * 16, U = UJ * 16;
    const int s = blockIdx.y;
    const int64_t r_lo = seg ? (int64_t)seg[s] * F : 0, r_hi = seg ? (int64_t)seg[s + 1] * F : M;
    constexpr int ROWS = 128 * WV;
    const int64_t row0 = r_lo + (int64_t)blockIdx.x * ROWS;
    if (row0 >= r_hi) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int RS1 = g2_row_slots(D) * 4, PL1 = g2_plane_floats(D, U);      // W1^T: K = D, N = U
    const int RS2 = g2_row_slots(U) * 4, PL2 = g2_plane_floats(U, D);      // W2^T: K = U, N = D
    float* img1 = g2_lds;
    float* img2 = g2_lds + 4 * PL1;
    const float* __restrict__ W1 = tab + (size_t)s * tab_stride;            // [D][U]
    const float* __restrict__ W2 = W1 + (size_t)D * U;                      // [U][D]
    g2_stage<false>(W1, D, U, U, img1, RS1, PL1);          // B(k = i, nn = u) = W1[i][u]
    g2_stage<false>(W2, U, D, D, img2, RS2, PL2);          // B(k = u, nn = o) = W2[u][o]
    __syncthreads();
    const float* b1 = img1 + g * PL1 + n * RS1;
    const float* b2 = img2 + g * PL2 + n * RS2;
    float4 a[KJ], an[KJ];
    auto load_rows = [&](int64_t g0, float4 (&dst)[KJ]) {
        const int64_t row = g0 + n;
        if (row < r_hi) {
            const float4* src = reinterpret_cast<const float4*>(z + row * D + 4 * g);
#pragma unroll
            for (int j = 0; j < KJ; ++j) dst[j] = src[4 * j];
        } else {
#pragma unroll
            for (int j = 0; j < KJ; ++j) dst[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int64_t wave_row0 = row0 + 16 * wave;
    const int64_t tile_hi = min(r_hi, row0 + ROWS);
    if (wave_row0 < tile_hi) load_rows(wave_row0, a);
    for (int64_t g0 = wave_row0; g0 < tile_hi; g0 += 16 * WV) {
        const int64_t row = g0 + n;
        const bool live = row < r_hi;
        // token (sample position, field) of this lane's row and its dropout key: the sample id is loaded BEFORE the next group's
        // rows are requested, so that waiting for it does not wait for them (loads return in order)
        const int64_t rr = live ? row : r_lo;
        const int pos = (int)(rr / F), f = (int)(rr - (int64_t)pos * F);
        const uint32_t skey = dc.on ? drop_sample_key(dc.key, (uint32_t)order[pos]) : 0u;
        const bool more = g0 + 16 * WV < tile_hi;
        if (more) load_rows(g0 + 16 * WV, an);
        float4 h[UJ];
#pragma unroll
        for (int ju = 0; ju < UJ; ++ju) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* bp = b1 + 16 * ju * RS1;
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const float4 b = *reinterpret_cast<const float4*>(bp + 4 * j);
                acc = mfma4(b.x, a[j].x, acc);
                acc = mfma4(b.y, a[j].y, acc);
                acc = mfma4(b.z, a[j].z, acc);
                acc = mfma4(b.w, a[j].w, acc);
            }
            h[ju] = make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
            if (live && hbuf) *reinterpret_cast<float4*>(hbuf + row * U + 16 * ju + 4 * g) = h[ju];
            __builtin_amdgcn_sched_barrier(0);      // the fragment reads of later tiles stay behind this tile (registers)
        }
        float v[KJ][4];
        float sum = 0.f;
#pragma unroll
        for (int jn = 0; jn < KJ; ++jn) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* bp = b2 + 16 * jn * RS2;
#pragma unroll
            for (int ju = 0; ju < UJ; ++ju) {
                const float4 b = *reinterpret_cast<const float4*>(bp + 4 * ju);
                acc = mfma4(b.x, h[ju].x, acc);
                acc = mfma4(b.y, h[ju].y, acc);
                acc = mfma4(b.z, h[ju].z, acc);
                acc = mfma4(b.w, h[ju].w, acc);
            }
            const int c = 16 * jn + 4 * g;
            const uint32_t kb = dc.on ? drop_keep4(skey, (uint32_t)(f * D + c) >> 2, dc.thresh) : 0xFu;
            const float zr[4] = {a[jn].x, a[jn].y, a[jn].z, a[jn].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[e];
                if (dc.on) x = (kb >> e) & 1u ? x * dc.scale : 0.f;
                x += zr[e];
                v[jn][e] = x;
                sum += x;
            }
            if (live && tbuf) *reinterpret_cast<float4*>(tbuf + row * D + c) = make_float4(v[jn][0], v[jn][1], v[jn][2], v[jn][3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int jn = 0; jn < KJ; ++jn)
#pragma unroll
            for (int e = 0; e < 4; ++e) q = fmaf(v[jn][e] - mean, v[jn][e] - mean, q);
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.0f / sqrtf(q * (1.0f / D) + 1e-6f);
        if (live) {
#pragma unroll
            for (int jn = 0; jn < KJ; ++jn) {
                const int c = 16 * jn + 4 * g;
                float4 o = make_float4(v[jn][0], v[jn][1], v[jn][2], v[jn][3]);
                if (norm) {
                    const float4 gm = *reinterpret_cast<const float4*>(gamma + c), bt = *reinterpret_cast<const float4*>(beta + c);
                    o = make_float4((o.x - mean) * rstd * gm.x + bt.x, (o.y - mean) * rstd * gm.y + bt.y,
                                    (o.z - mean) * rstd * gm.z + bt.z, (o.w - mean) * rstd * gm.w + bt.w);
                }
                *reinterpret_cast<float4*>(y + row * D + c) = o;
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < KJ; ++j) a[j] = an[j];
        }
    }
}

// ---- Output block forward in one pass (satrans.py:91-99): u = o Wo^T, t = drop(relu?(u)) + x, y = LayerNorm(t) -----------------------
// The product of gen_gemm_kernel<true, 0> with the LayerNorm launch as its epilogue: a token's D = 16 KJ outputs sit in the
// four lanes g of one wave (as in gen_metanet_fwd_kernel), so the statistics are two cross-lane adds.  u is written only when the
// backward needs it (flag `relu`: the mask [u > 0]); t (saved for the backward) and y always; y in the caller's sample order
// when `y_orig`.  Reads o and x once, writes t and y: 537 MB at the configs[4] shape where the two launches move 940 MB.
template <int KJ>
__global__ __launch_bounds__(256) void gen_out_ln_fwd_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                           const float* __restrict__ res, float* __restrict__ u_out,
                                                           float* __restrict__ t_out, float* __restrict__ y, bool y_orig,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int M, int F, const int32_t* __restrict__ order, GenDrop dc, bool relu) {
    extern __shared__ __align__(16) float g2_lds[];
    constexpr int D = KJ * 16;
    const int64_t row0 = (int64_t)blockIdx.x * kG2Rows;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int RS = g2_row_slots(D) * 4, PL = g2_plane_floats(D, D);
    g2_stage<true>(W, D, D, D, g2_lds, RS, PL);            // B(k, nn) = Wo[nn][k]
    __syncthreads();
    const float* bt = g2_lds + g * PL + n * RS;
    float4 a[KJ], an[KJ];
    auto load_rows = [&](int64_t g0, float4 (&dst)[KJ]) {
        const int64_t row = g0 + n;
        if (row < M) {
            const float4* src = reinterpret_cast<const float4*>(A + row * D + 4 * g);
#pragma unroll
            for (int j = 0; j < KJ; ++j) dst[j] = src[4 * j];
        } else {
#pragma unroll
            for (int j = 0; j < KJ; ++j) dst[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int64_t wave_row0 = row0 + 16 * wave;
    const int64_t tile_hi = min((int64_t)M, row0 + kG2Rows);
    if (wave_row0 < tile_hi) load_rows(wave_row0, a);
    for (int64_t g0 = wave_row0; g0 < tile_hi; g0 += 64) {
        const int64_t row = g0 + n;
        const bool live = row < M;
        const int64_t rr = live ? row : 0;
        const int pos = (int)(rr / F), f = (int)(rr - (int64_t)pos * F);
        const int b = order[pos];      // (loaded BEFORE the next group's rows are requested: loads return in order)
        float4 rs[KJ];
#pragma unroll
        for (int jn = 0; jn < KJ; ++jn)
            rs[jn] = res ? *reinterpret_cast<const float4*>(res + rr * D + 16 * jn + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        const bool more = g0 + 64 < tile_hi;
        if (more) load_rows(g0 + 64, an);
        const uint32_t skey = dc.on ? drop_sample_key(dc.key, (uint32_t)b) : 0u;
        float v[KJ][4];
        float sum = 0.f;
#pragma unroll
        for (int jn = 0; jn < KJ; ++jn) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* bp = bt + 16 * jn * RS;
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const float4 bb = *reinterpret_cast<const float4*>(bp + 4 * j);
                acc = mfma4(bb.x, a[j].x, acc);
                acc = mfma4(bb.y, a[j].y, acc);
                acc = mfma4(bb.z, a[j].z, acc);
                acc = mfma4(bb.w, a[j].w, acc);
            }
            const int c = 16 * jn + 4 * g;
            if (live && u_out) *reinterpret_cast<float4*>(u_out + row * D + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            const uint32_t kb = dc.on ? drop_keep4(skey, (uint32_t)(f * D + c) >> 2, dc.thresh) : 0xFu;
            const float zr[4] = {rs[jn].x, rs[jn].y, rs[jn].z, rs[jn].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[e];
                if (relu) x = fmaxf(x, 0.f);
                if (dc.on) x = (kb >> e) & 1u ? x * dc.scale : 0.f;
                x += zr[e];
                v[jn][e] = x;
                sum += x;
            }
            if (live) *reinterpret_cast<float4*>(t_out + row * D + c) = make_float4(v[jn][0], v[jn][1], v[jn][2], v[jn][3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int jn = 0; jn < KJ; ++jn)
#pragma unroll
            for (int e = 0; e < 4; ++e) q = fmaf(v[jn][e] - mean, v[jn][e] - mean, q);
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.0f / sqrtf(q * (1.0f / D) + 1e-6f);
        if (live) {
            const int64_t dst = y_orig ? (int64_t)b * F + f : row;
#pragma unroll
            for (int jn = 0; jn < KJ; ++jn) {
                const int c = 16 * jn + 4 * g;
                const float4 gm = *reinterpret_cast<const float4*>(gamma + c), bt4 = *reinterpret_cast<const float4*>(beta + c);
                *reinterpret_cast<float4*>(y + dst * D + c) =
                    make_float4((v[jn][0] - mean) * rstd * gm.x + bt4.x, (v[jn][1] - mean) * rstd * gm.y + bt4.y,
                                (v[jn][2] - mean) * rstd * gm.z + bt4.z, (v[jn][3] - mean) * rstd * gm.w + bt4.w);
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < KJ; ++j) a[j] = an[j];
        }
    }
}

// ---- MetaNet backward, the data-gradient chain in one pass: LayerNorm backward -> dm = mask(dt) -> dh = (dm W2^T) [h > 0] ->
// dz = dt + dh W1^T, with the same chaining as the forward kernel (dm tiles are the second operand of the contraction over D, dh
// tiles of the contraction over U).  dm and dh are written out because the weight-gradient products (dW2 = h^T dm, dW1 = z^T dh)
// read them; g is overwritten with dz.  Reads g, t, h once (536 MB at the configs[4] shape), writes dm, dh, dz (536 MB) where the
// three separate launches move 1.74 GB.  LayerNorm gamma / beta gradient partials per workgroup into `part` ([gridDim.x * S][2 D],
// every workgroup writes its slot - zeros when it holds no row), summed in a fixed order by gen_tn_reduce_kernel.
template <int KJ, int UJ>
__global__ __launch_bounds__(256, 2) void gen_metanet_bwd_kernel(float* __restrict__ g, const float* __restrict__ t,
                                                               const float* __restrict__ hbuf, const float* __restrict__ tab,
                                                               int64_t tab_stride, const int32_t* __restrict__ seg, int M, int F,
                                                               float* __restrict__ dm_out, float* __restrict__ dh_out,
                                                               const float* __restrict__ gamma, float* __restrict__ part,
                                                               const int32_t* __restrict__ order, GenDrop dc) {
    extern __shared__ __align__(16) float g2_lds[];
    constexpr int D = KJ * 16, U = UJ * 16;
    const int s = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g_ = lane >> 4;
    float* my_part = part + ((size_t)s * gridDim.x + blockIdx.x) * 2 * D;
    const int64_t r_lo = seg ? (int64_t)seg[s] * F : 0, r_hi = seg ? (int64_t)seg[s + 1] * F : M;
    const int64_t row0 = r_lo + (int64_t)blockIdx.x * kG2Rows;
    if (row0 >= r_hi) {
        for (int i = tid; i < 2 * D; i += 256) my_part[i] = 0.f;
        return;
    }
    const int RS2 = g2_row_slots(D) * 4, PL2 = g2_plane_floats(D, U);      // dh = dm W2^T: K = D, N = U, B(k = o, n = u) = W2[u][o]
    const int RS1 = g2_row_slots(U) * 4, PL1 = g2_plane_floats(U, D);      // dz = dh W1^T: K = U, N = D, B(k = u, n = i) = W1[i][u]
    float* img2 = g2_lds;
    float* img1 = g2_lds + 4 * PL2;
    const float* __restrict__ W1 = tab + (size_t)s * tab_stride;            // [D][U]
    const float* __restrict__ W2 = W1 + (size_t)D * U;                      // [U][D]
    g2_stage<true>(W2, D, U, D, img2, RS2, PL2);           // B(k = o, nn = u) = W2[u][o]
    g2_stage<true>(W1, U, D, U, img1, RS1, PL1);           // B(k = u, nn = i) = W1[i][u]
    __syncthreads();
    const float* b2 = img2 + g_ * PL2 + n * RS2;
    const float* b1 = img1 + g_ * PL1 + n * RS1;
    float gmv[KJ][4];
#pragma unroll
    for (int jn = 0; jn < KJ; ++jn) {
        const float4 gm4 = *reinterpret_cast<const float4*>(gamma + 16 * jn + 4 * g_);
        gmv[jn][0] = gm4.x; gmv[jn][1] = gm4.y; gmv[jn][2] = gm4.z; gmv[jn][3] = gm4.w;
    }
    float ag[KJ][4], ab[KJ][4];
#pragma unroll
    for (int jn = 0; jn < KJ; ++jn)
#pragma unroll
        for (int e = 0; e < 4; ++e) { ag[jn][e] = 0.f; ab[jn][e] = 0.f; }
    const int64_t wave_row0 = row0 + 16 * wave;
    const int64_t tile_hi = min(r_hi, row0 + kG2Rows);
    for (int64_t g0 = wave_row0; g0 < tile_hi; g0 += 64) {
        const int64_t row = g0 + n;
        const bool live = row < r_hi;
        const int64_t rr = live ? row : r_lo;
        float gv[KJ][4], zh[KJ][4];
        float sum = 0.f;
        float4 hv[UJ];                    // the relu mask rows: requested now, used after the LayerNorm arithmetic
#pragma unroll
        for (int ju = 0; ju < UJ; ++ju) hv[ju] = *reinterpret_cast<const float4*>(hbuf + rr * U + 16 * ju + 4 * g_);
#pragma unroll
        for (int jn = 0; jn < KJ; ++jn) {
            const int c = 16 * jn + 4 * g_;
            const float4 g4 = *reinterpret_cast<const float4*>(g + rr * D + c);
            const float4 t4 = *reinterpret_cast<const float4*>(t + rr * D + c);
            gv[jn][0] = live ? g4.x : 0.f; gv[jn][1] = live ? g4.y : 0.f; gv[jn][2] = live ? g4.z : 0.f; gv[jn][3] = live ? g4.w : 0.f;
            zh[jn][0] = t4.x; zh[jn][1] = t4.y; zh[jn][2] = t4.z; zh[jn][3] = t4.w;
            sum += (t4.x + t4.y) + (t4.z + t4.w);
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int jn = 0; jn < KJ; ++jn)
#pragma unroll
            for (int e = 0; e < 4; ++e) q = fmaf(zh[jn][e] - mean, zh[jn][e] - mean, q);
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.0f / sqrtf(q * (1.0f / D) + 1e-6f);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int jn = 0; jn < KJ; ++jn)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                zh[jn][e] = (zh[jn][e] - mean) * rstd;
                ag[jn][e] = fmaf(gv[jn][e], zh[jn][e], ag[jn][e]);
                ab[jn][e] += gv[jn][e];
                gv[jn][e] *= gmv[jn][e];
                m1 += gv[jn][e];
                m2 = fmaf(gv[jn][e], zh[jn][e], m2);
            }
        m1 += __shfl_xor(m1, 16, 64); m1 += __shfl_xor(m1, 32, 64);
        m2 += __shfl_xor(m2, 16, 64); m2 += __shfl_xor(m2, 32, 64);
        m1 *= (1.0f / D);
        m2 *= (1.0f / D);
        const int pos = (int)(rr / F), f = (int)(rr - (int64_t)pos * F);
        const uint32_t skey = dc.on ? drop_sample_key(dc.key, (uint32_t)order[pos]) : 0u;
        float4 dmt[KJ];                    // dm tiles: second operand of dh = dm W2^T; gv becomes dt
#pragma unroll
        for (int jn = 0; jn < KJ; ++jn) {
            const int c = 16 * jn + 4 * g_;
            const uint32_t kb = dc.on ? drop_keep4(skey, (uint32_t)(f * D + c) >> 2, dc.thresh) : 0xFu;
            float dmv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d_ = rstd * (gv[jn][e] - m1 - zh[jn][e] * m2);
                gv[jn][e] = d_;
                dmv[e] = dc.on ? ((kb >> e) & 1u ? d_ * dc.scale : 0.f) : d_;
            }
            dmt[jn] = make_float4(dmv[0], dmv[1], dmv[2], dmv[3]);
            if (live) *reinterpret_cast<float4*>(dm_out + row * D + c) = dmt[jn];
        }
        float4 dh[UJ];
#pragma unroll
        for (int ju = 0; ju < UJ; ++ju) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* bp = b2 + 16 * ju * RS2;
#pragma unroll
            for (int jn = 0; jn < KJ; ++jn) {
                const float4 b = *reinterpret_cast<const float4*>(bp + 4 * jn);
                acc = mfma4(b.x, dmt[jn].x, acc);
                acc = mfma4(b.y, dmt[jn].y, acc);
                acc = mfma4(b.z, dmt[jn].z, acc);
                acc = mfma4(b.w, dmt[jn].w, acc);
            }
            dh[ju] = make_float4(hv[ju].x > 0.f ? acc[0] : 0.f, hv[ju].y > 0.f ? acc[1] : 0.f, hv[ju].z > 0.f ? acc[2] : 0.f,
                                 hv[ju].w > 0.f ? acc[3] : 0.f);
            if (live) *reinterpret_cast<float4*>(dh_out + row * U + 16 * ju + 4 * g_) = dh[ju];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int jn = 0; jn < KJ; ++jn) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* bp = b1 + 16 * jn * RS1;
#pragma unroll
            for (int ju = 0; ju < UJ; ++ju) {
                const float4 b = *reinterpret_cast<const float4*>(bp + 4 * ju);
                acc = mfma4(b.x, dh[ju].x, acc);
                acc = mfma4(b.y, dh[ju].y, acc);
                acc = mfma4(b.z, dh[ju].z, acc);
                acc = mfma4(b.w, dh[ju].w, acc);
            }
            if (live)
                *reinterpret_cast<float4*>(g + row * D + 16 * jn + 4 * g_) =
                    make_float4(acc[0] + gv[jn][0], acc[1] + gv[jn][1], acc[2] + gv[jn][2], acc[3] + gv[jn][3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // gamma / beta partials of this workgroup: column c of the 16 token lanes x 4 waves, summed in a fixed order
    __syncthreads();                                   // (the weight images are dead: the LDS is reused)
    float* red = g2_lds;                               // [2][64 rows = wave * 16 + n][D]
#pragma unroll
    for (int jn = 0; jn < KJ; ++jn)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[(size_t)(wave * 16 + n) * D + 16 * jn + 4 * g_ + e] = ag[jn][e];
            red[(size_t)(64 + wave * 16 + n) * D + 16 * jn + 4 * g_ + e] = ab[jn][e];
        }
    __syncthreads();
    for (int i = tid; i < 2 * D; i += 256) {
        const int which = i / D, col = i - which * D;
        float s_ = 0.f;
        for (int k = 0; k < 64; ++k) s_ += red[(size_t)(which * 64 + k) * D + col];
        my_part[i] = s_;
    }
}

// t = drop(relu?(a)) + res ; y = LayerNorm(t) * gamma + beta       (submodules.py:96-101, satrans.py:91-99)
// LPT = D / 4 lanes per token.  y goes to the caller's sample order when `y_orig` is set.
template <int LPT>
__global__ __launch_bounds__(256) void gen_ln_fwd_kernel(const float* __restrict__ a, const float* __restrict__ res,
                                                       float* __restrict__ t_out, float* __restrict__ y, bool y_orig,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int64_t Ntok, int F, const int32_t* __restrict__ order, GenDrop dc,
                                                       bool relu, bool relu_post, bool norm) {
    // relu_post (SelfAttention_Layer, submodules.py:230-235): t = drop(a) + res is SAVED, the norm sees relu(t);
    // !norm (MetaNet(use_norm=False), submodules.py:100-102): y = t
    constexpr int D = 4 * LPT;
    const int64_t m = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPT;
    const int c = (int)(threadIdx.x % LPT) * 4;
    const bool live = m < Ntok;
    const int64_t mm = live ? m : 0;
    const int p = (int)(mm / F), f = (int)(mm - (int64_t)p * F);
    const int b = order[p];
    float4 v4 = *reinterpret_cast<const float4*>(a + mm * D + c);
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
    const uint32_t kb = dc.on ? drop_keep4(drop_sample_key(dc.key, (uint32_t)b), (uint32_t)(f * D + c) >> 2, dc.thresh) : 0xFu;
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (res) {
        const float4 r4 = *reinterpret_cast<const float4*>(res + mm * D + c);
        rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w;
    }
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e];
        if (relu) x = fmaxf(x, 0.f);
        if (dc.on) x = (kb >> e) & 1u ? x * dc.scale : 0.f;
        x += rs[e];
        v[e] = x;
    }
    if (live && t_out) *reinterpret_cast<float4*>(t_out + m * D + c) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (relu_post) v[e] = fmaxf(v[e], 0.f);
        sum += v[e];
    }
#pragma unroll
    for (int o = 1; o < LPT; o <<= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) q = fmaf(v[e] - mean, v[e] - mean, q);
#pragma unroll
    for (int o = 1; o < LPT; o <<= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / D) + 1e-6f);
    if (!live) return;
    const int64_t dst = y_orig ? (int64_t)b * F + f : m;
    if (!norm) {
        *reinterpret_cast<float4*>(y + dst * D + c) = make_float4(v[0], v[1], v[2], v[3]);
        return;
    }
    const float4 gm = *reinterpret_cast<const float4*>(gamma + c), bt = *reinterpret_cast<const float4*>(beta + c);
    *reinterpret_cast<float4*>(y + dst * D + c) = make_float4((v[0] - mean) * rstd * gm.x + bt.x, (v[1] - mean) * rstd * gm.y + bt.y,
                                                              (v[2] - mean) * rstd * gm.z + bt.z, (v[3] - mean) * rstd * gm.w + bt.w);
}

// LayerNorm backward + the mask of the branch in front of it.  g: gradient of the LayerNorm output (in the caller's sample
// order when g_orig), t: saved pre-norm rows.  dt: gradient of t (= of the residual input); dm = dt * dropout mask
// (* [a > 0] with relu).  gamma / beta gradients: per-block partial sums [blocks][2 D], blocks take contiguous token ranges.
template <int LPT>
__global__ __launch_bounds__(256) void gen_ln_bwd_kernel(const float* __restrict__ g, bool g_orig, const float* __restrict__ t,
                                                       const float* __restrict__ a_pre, const float* __restrict__ gamma,
                                                       float* __restrict__ dt, float* __restrict__ dm, float* __restrict__ part,
                                                       int64_t Ntok, int F, const int32_t* __restrict__ order, GenDrop dc,
                                                       bool relu, int tokens_per_block, bool relu_post, bool norm) {
    constexpr int D = 4 * LPT, TPI = 256 / LPT;      // tokens per iteration
    __shared__ float red[2][TPI][D];
    const int sub = threadIdx.x / LPT, c = (int)(threadIdx.x % LPT) * 4;
    const int64_t t_lo = (int64_t)blockIdx.x * tokens_per_block, t_hi = min(Ntok, t_lo + tokens_per_block);
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    float gm[4] = {1.f, 1.f, 1.f, 1.f};
    if (norm) {
        const float4 gm4 = *reinterpret_cast<const float4*>(gamma + c);
        gm[0] = gm4.x; gm[1] = gm4.y; gm[2] = gm4.z; gm[3] = gm4.w;
    }
    for (int64_t base = t_lo; base < t_hi; base += TPI) {
        const int64_t m = base + sub;
        const bool live = m < t_hi;
        const int64_t mm = live ? m : t_lo;
        const int p = (int)(mm / F), f = (int)(mm - (int64_t)p * F);
        const int b = order[p];
        const int64_t src = g_orig ? (int64_t)b * F + f : mm;
        const float4 g4 = *reinterpret_cast<const float4*>(g + src * D + c);
        const float4 t4 = *reinterpret_cast<const float4*>(t + mm * D + c);
        const float g4_raw[4] = {g4.x, g4.y, g4.z, g4.w};
        float gv[4] = {g4.x, g4.y, g4.z, g4.w}, tv[4] = {t4.x, t4.y, t4.z, t4.w};
        if (!live) { gv[0] = gv[1] = gv[2] = gv[3] = 0.f; }
        bool pos_[4] = {true, true, true, true};            // relu_post: the norm saw relu(t); its gradient passes where t > 0
        if (relu_post) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { pos_[e] = tv[e] > 0.f; tv[e] = fmaxf(tv[e], 0.f); }
        }
        float sum = (tv[0] + tv[1]) + (tv[2] + tv[3]);
#pragma unroll
        for (int o = 1; o < LPT; o <<= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) q = fmaf(tv[e] - mean, tv[e] - mean, q);
#pragma unroll
        for (int o = 1; o < LPT; o <<= 1) q += __shfl_xor(q, o, 64);
        const float rstd = 1.0f / sqrtf(q * (1.0f / D) + 1e-6f);
        float zh[4], m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            zh[e] = (tv[e] - mean) * rstd;
            ag[e] = fmaf(gv[e], zh[e], ag[e]);
            ab[e] += gv[e];
            gv[e] *= gm[e];
            m1 += gv[e];
            m2 = fmaf(gv[e], zh[e], m2);
        }
#pragma unroll
        for (int o = 1; o < LPT; o <<= 1) { m1 += __shfl_xor(m1, o, 64); m2 += __shfl_xor(m2, o, 64); }
        m1 *= (1.0f / D);
        m2 *= (1.0f / D);
        if (live) {
            float d_[4], dmv[4];
            const uint32_t kb = dc.on ? drop_keep4(drop_sample_key(dc.key, (uint32_t)b), (uint32_t)(f * D + c) >> 2, dc.thresh) : 0xFu;
            float pre[4] = {1.f, 1.f, 1.f, 1.f};
            if (relu) {
                const float4 p4 = *reinterpret_cast<const float4*>(a_pre + m * D + c);
                pre[0] = p4.x; pre[1] = p4.y; pre[2] = p4.z; pre[3] = p4.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d_[e] = norm ? rstd * (gv[e] - m1 - zh[e] * m2) : g4_raw[e];
                if (!pos_[e]) d_[e] = 0.f;
                float mk = 1.0f;
                if (dc.on) mk = (kb >> e) & 1u ? dc.scale : 0.f;
                if (relu && !(pre[e] > 0.f)) mk = 0.f;
                dmv[e] = d_[e] * mk;
            }
            if (dt) *reinterpret_cast<float4*>(dt + m * D + c) = make_float4(d_[0], d_[1], d_[2], d_[3]);
            *reinterpret_cast<float4*>(dm + m * D + c) = make_float4(dmv[0], dmv[1], dmv[2], dmv[3]);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][sub][c + e] = ag[e]; red[1][sub][c + e] = ab[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        const int which = i / D, col = i - which * D;
        float s_ = 0.f;
        for (int k = 0; k < TPI; ++k) s_ += red[which][k][col];
        part[(size_t)blockIdx.x * 2 * D + i] = s_;
    }
}

// ---- attention, one lane per (head, query row) of one sample per workgroup: the "wavefront" arm --------------------------------
// q, k, v: [N, D] sorted rows; o: [N, D]; st: [B, H, F] float2 (max of the scaled scores, 1 / sum); att: optional [H,B,F,F]
template <int d>
__global__ __launch_bounds__(256) void gen_attn_fwd_wave_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, float* __restrict__ o,
                                                              float2* __restrict__ st, float* __restrict__ att, int B, int F,
                                                              int H, const int32_t* __restrict__ order, GenDrop dc,
                                                              float inv_sqrt_d) {
    extern __shared__ __align__(16) float at_lds[];
    const int D = H * d, LD = D + 4;
    float* sk = at_lds;
    float* sv = at_lds + F * LD;
    const int p = blockIdx.x;
    const int64_t base = (int64_t)p * F * D;
    for (int i = threadIdx.x; i < F * (D >> 2); i += blockDim.x) {
        const int r = i / (D >> 2), c = (i - r * (D >> 2)) * 4;
        *reinterpret_cast<float4*>(sk + r * LD + c) = *reinterpret_cast<const float4*>(k + base + r * D + c);
        *reinterpret_cast<float4*>(sv + r * LD + c) = *reinterpret_cast<const float4*>(v + base + r * D + c);
    }
    __syncthreads();
    const int b = order[p];
    const float sc_scale = kLog2e * inv_sqrt_d;
    for (int task = threadIdx.x; task < H * F; task += blockDim.x) {
        const int h = task / F, i = task - h * F;
        f32x2 qi[d / 2];
        load_row<d>(q + base + (int64_t)i * D + h * d, qi);
        const float* kb_ = sk + h * d;
        const float* vb_ = sv + h * d;
        float mx = -INFINITY;
#pragma unroll 1
        for (int j0 = 0; j0 < F; j0 += 4) {
            f32x2 kr[4][d / 2];
#pragma unroll
            for (int u = 0; u < 4; ++u) load_row<d>(kb_ + min(j0 + u, F - 1) * LD, kr[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) mx = fmaxf(mx, dot_row<d>(qi, kr[u]) * sc_scale);
        }
        f32x2 oacc[d / 2];
#pragma unroll
        for (int e = 0; e < d / 2; ++e) oacc[e] = f32x2{0.f, 0.f};
        float sum = 0.f;
        const uint32_t skey = drop_sample_key(dc.key, (uint32_t)b);
        const uint32_t block0 = drop_attn_elem(h, F, i, 0) >> 2;
#pragma unroll 1
        for (int j0 = 0; j0 < F; j0 += 4) {
            f32x2 kr[4][d / 2], vr[4][d / 2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                load_row<d>(kb_ + min(j0 + u, F - 1) * LD, kr[u]);
                load_row<d>(vb_ + min(j0 + u, F - 1) * LD, vr[u]);
            }
            const uint32_t kb = dc.on ? drop_keep4(skey, block0 + (uint32_t)(j0 >> 2), dc.thresh) : 0xFu;
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
        st[((size_t)p * H + h) * F + i] = make_float2(mx, inv);
        float orow[d];
#pragma unroll
        for (int e = 0; e < d / 2; ++e) { orow[2 * e] = oacc[e].x * inv; orow[2 * e + 1] = oacc[e].y * inv; }
#pragma unroll
        for (int e = 0; e < d; e += 4)
            *reinterpret_cast<float4*>(o + base + (int64_t)i * D + h * d + e) = make_float4(orow[e], orow[e + 1], orow[e + 2], orow[e + 3]);
        if (att) {      // normalized_att_scores [H,B,F,F] after dropout (satrans.py:87), in the caller's sample order
            float* arow = att + (((size_t)h * B + b) * F + i) * F;
            for (int j = 0; j < F; ++j) {
                f32x2 kr[d / 2];
                load_row<d>(kb_ + j * LD, kr);
                float pj = __builtin_amdgcn_exp2f(dot_row<d>(qi, kr) * sc_scale - mx) * inv;
                if (dc.on) pj = drop_keep(skey, drop_attn_elem(h, F, i, j), dc.thresh) ? pj * dc.scale : 0.f;
                arow[j] = pj;
            }
        }
    }
}

// ---- attention forward on the matrix pipe: F <= 64 keys, head dimension 16, one wave per head, one sample per workgroup ----------
// Scores are computed TRANSPOSED, S^T[j][i] = sum_e K[j][e] Q[i][e]: the query sits on the MFMA column (= lane & 15), the keys in
// the accumulator rows (4 key tiles x 4 registers per lane, the rest across the four lane groups), so the softmax over the keys
// is a register reduction plus two shuffles, and P^T is - as it stands in the accumulators - the B operand of
// O^T[e][i] = sum_j V[j][e] P^T[j][i]: nothing F x F ever leaves the registers.  O^T comes out in the token-on-lane layout.
__global__ __launch_bounds__(256) void gen_attn_fwd_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, float* __restrict__ o,
                                                              float2* __restrict__ st, int F, int H,
                                                              const int32_t* __restrict__ order, GenDrop dc, float inv_sqrt_d) {
    constexpr int d = 16;
    extern __shared__ __align__(16) float at_lds[];
    const int D = H * d, LD = D + 4;
    const int FP = (F + 15) & ~15;
    float* sq = at_lds;
    float* sk = sq + FP * LD;
    float* sv = sk + FP * LD;
    const int p = blockIdx.x;
    const int64_t base = (int64_t)p * F * D;
    for (int i = threadIdx.x; i < FP * (D >> 2); i += blockDim.x) {
        const int r = i / (D >> 2), c = (i - r * (D >> 2)) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, cc = a;
        if (r < F) {
            a = *reinterpret_cast<const float4*>(q + base + r * D + c);
            b = *reinterpret_cast<const float4*>(k + base + r * D + c);
            cc = *reinterpret_cast<const float4*>(v + base + r * D + c);
        }
        *reinterpret_cast<float4*>(sq + r * LD + c) = a;
        *reinterpret_cast<float4*>(sk + r * LD + c) = b;
        *reinterpret_cast<float4*>(sv + r * LD + c) = cc;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    if (h >= H) return;
    const int b = order[p];
    const float sc_scale = kLog2e * inv_sqrt_d;
    const int JT = FP >> 4;
    const uint32_t skey = drop_sample_key(dc.key, (uint32_t)b);
    for (int it = 0; it < JT; ++it) {               // query tile: queries 16 it + n on the lanes
        const int i = 16 * it + n;
        float qb[4];                                // B operand: Q^T[e = 4 ks + g][i]
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qb[ks] = sq[i * LD + h * d + 4 * ks + g];
        f32x4 s[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            s[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (jt < JT) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)      // A operand: K[j = 16 jt + n][e = 4 ks + g]
                    s[jt] = mfma4(sk[(16 * jt + n) * LD + h * d + 4 * ks + g], qb[ks], s[jt]);
            }
        }
        // lane (n, g), register r of tile jt holds key j = 16 jt + 4 g + r for query i
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool real = 16 * jt + 4 * g + r < F;
                s[jt][r] = real ? s[jt][r] * sc_scale : -INFINITY;
                mx = fmaxf(mx, s[jt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
        const uint32_t block0 = drop_attn_elem(h, F, i < F ? i : 0, 0) >> 2;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const uint32_t kb = dc.on ? drop_keep4(skey, block0 + (uint32_t)(4 * jt + g), dc.thresh) : 0xFu;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ex = __builtin_amdgcn_exp2f(s[jt][r] - mx);     // exp2(-inf) = 0 for padding keys
                sum += ex;
                s[jt][r] = dc.on ? ((kb >> r) & 1u ? ex * dc.scale : 0.f) : ex;
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        f32x4 ot = f32x4{0.f, 0.f, 0.f, 0.f};       // O^T[e = 4 g + r][i]
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
            if (jt < JT) {
#pragma unroll
                for (int r = 0; r < 4; ++r)         // contraction step over keys j = 16 jt + 4 g + r: A = V[j][e = n]
                    ot = mfma4(sv[(16 * jt + 4 * g + r) * LD + h * d + n], s[jt][r], ot);
            }
        if (i < F) {
            *reinterpret_cast<float4*>(o + base + (int64_t)i * D + h * d + 4 * g) =
                make_float4(ot[0] * inv, ot[1] * inv, ot[2] * inv, ot[3] * inv);
            if (g == 0) st[((size_t)p * H + h) * F + i] = make_float2(mx, inv);
        }
    }
}

// ---- attention backward, one lane per (head, row) of one sample per workgroup: dq_i (row pass) and dk_i, dv_i (column pass)
// with P recomputed from q, k and the saved row statistics; dot_i = go_i . o_i          (the recomputing scheme of round 2's 8-wave fused backward)
template <int d>
__global__ __launch_bounds__(256) void gen_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ o,
                                                         const float* __restrict__ go, const float2* __restrict__ st,
                                                         float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                         int F, int H, const int32_t* __restrict__ order, GenDrop dc,
                                                         float inv_sqrt_d) {
    extern __shared__ __align__(16) float at_lds[];
    const int D = H * d, LD = D + 4;
    float* sq = at_lds;
    float* sk = sq + F * LD;
    float* sv = sk + F * LD;
    float* sg = sv + F * LD;
    float4* sst = reinterpret_cast<float4*>(sg + F * LD);      // [H * F]: max, 1/sum, dot, keep bits low / high
    const int p = blockIdx.x;
    const int64_t base = (int64_t)p * F * D;
    for (int i = threadIdx.x; i < F * (D >> 2); i += blockDim.x) {
        const int r = i / (D >> 2), c = (i - r * (D >> 2)) * 4;
        *reinterpret_cast<float4*>(sq + r * LD + c) = *reinterpret_cast<const float4*>(q + base + r * D + c);
        *reinterpret_cast<float4*>(sk + r * LD + c) = *reinterpret_cast<const float4*>(k + base + r * D + c);
        *reinterpret_cast<float4*>(sv + r * LD + c) = *reinterpret_cast<const float4*>(v + base + r * D + c);
        *reinterpret_cast<float4*>(sg + r * LD + c) = *reinterpret_cast<const float4*>(go + base + r * D + c);
    }
    __syncthreads();
    const int b = order[p];
    const uint32_t skey = drop_sample_key(dc.key, (uint32_t)b);
    for (int task = threadIdx.x; task < H * F; task += blockDim.x) {
        const int h = task / F, i = task - h * F;
        f32x2 gi[d / 2], oi[d / 2];
        load_row<d>(sg + i * LD + h * d, gi);
        load_row<d>(o + base + (int64_t)i * D + h * d, oi);
        const float2 s2 = st[((size_t)p * H + h) * F + i];
        sst[task] = make_float4(s2.x, s2.y, dot_row<d>(gi, oi), 0.f);
    }
    __syncthreads();
    const float sc_scale = inv_sqrt_d * kLog2e, scale = dc.scale;
    for (int task = threadIdx.x; task < H * F; task += blockDim.x) {
        const int h = task / F, i = task - h * F;
        const float* qb_ = sq + h * d;
        const float* kb_ = sk + h * d;
        const float* vb_ = sv + h * d;
        const float* gb_ = sg + h * d;
        const float4* stb = sst + h * F;
        f32x2 dqa[d / 2], dka[d / 2], dva[d / 2];
#pragma unroll
        for (int e = 0; e < d / 2; ++e) { dqa[e] = f32x2{0.f, 0.f}; dka[e] = f32x2{0.f, 0.f}; dva[e] = f32x2{0.f, 0.f}; }
        {   // row i
            f32x2 qi[d / 2], gi[d / 2];
            load_row<d>(qb_ + i * LD, qi);
            load_row<d>(gb_ + i * LD, gi);
            const float4 s4 = stb[i];
#pragma unroll 1
            for (int j = 0; j < F; ++j) {
                f32x2 kr[d / 2], vr[d / 2];
                load_row<d>(kb_ + j * LD, kr);
                load_row<d>(vb_ + j * LD, vr);
                const bool kp = !dc.on || drop_keep(skey, drop_attn_elem(h, F, i, j), dc.thresh);
                const float dp = kp ? dot_row<d>(gi, vr) * scale : 0.f;
                const float pj = __builtin_amdgcn_exp2f(dot_row<d>(qi, kr) * sc_scale - s4.x) * s4.y;
                axpy_row<d>(pj * (dp - s4.z) * inv_sqrt_d, kr, dqa);
            }
        }
        {   // column i
            f32x2 ki[d / 2], vi[d / 2];
            load_row<d>(kb_ + i * LD, ki);
            load_row<d>(vb_ + i * LD, vi);
#pragma unroll 1
            for (int r = 0; r < F; ++r) {
                f32x2 qr[d / 2], gr[d / 2];
                load_row<d>(qb_ + r * LD, qr);
                load_row<d>(gb_ + r * LD, gr);
                const float4 sr = stb[r];
                const bool kp = !dc.on || drop_keep(skey, drop_attn_elem(h, F, r, i), dc.thresh);
                const float pr = __builtin_amdgcn_exp2f(dot_row<d>(qr, ki) * sc_scale - sr.x) * sr.y;
                const float dp = kp ? dot_row<d>(gr, vi) * scale : 0.f;
                axpy_row<d>(pr * (dp - sr.z) * inv_sqrt_d, qr, dka);
                axpy_row<d>(kp ? pr * scale : 0.f, gr, dva);
            }
        }
        const int64_t at = base + (int64_t)i * D + h * d;
#pragma unroll
        for (int e = 0; e < d / 2; e += 2) {
            *reinterpret_cast<float4*>(dq + at + 2 * e) = make_float4(dqa[e].x, dqa[e].y, dqa[e + 1].x, dqa[e + 1].y);
            *reinterpret_cast<float4*>(dk + at + 2 * e) = make_float4(dka[e].x, dka[e].y, dka[e + 1].x, dka[e + 1].y);
            *reinterpret_cast<float4*>(dv + at + 2 * e) = make_float4(dva[e].x, dva[e].y, dva[e + 1].x, dva[e + 1].y);
        }
    }
}

// ---- attention backward on the matrix pipe (d = 16, F <= 64, H <= 4): one workgroup per sample, one wave per head, the scores
// TRANSPOSED as in the forward kernel.  Per tile of 16 queries (query i on the lane's column n):
//   S^T  = K Q^T and dP^T = V G^T          lane (i, g), register r of tile jt <-> key j = 16 jt + 4 g + r
//   P = exp2(s - max_i) / sum_i from the forward's row statistics, keep bits regenerated, dot_i = go_i . o_i:
//   dS^T = P (keep dP scale - dot_i) / sqrt(d),  Pm^T = keep P scale                              - all in registers
//   dQ^T = K^T dS^T: the accumulator tiles ARE the second operand (contraction over the register index j)
//   dK^T += Q^T dS, dV^T += G^T Pm contract over the QUERIES, which sit on the lanes: the two tiles go through a wave-private
//   LDS scratch (written by rows j, read back 16 bytes along i) - 32 ds_write_b32 + 8 ds_read_b128 per query tile.
// Every output tile is transposed (lane = token, 4 consecutive features in its registers): 16-byte stores.
__global__ __launch_bounds__(256) void gen_attn_bwd_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const float* __restrict__ o,
                                                              const float* __restrict__ go, const float2* __restrict__ st,
                                                              float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                              int F, int H, const int32_t* __restrict__ order, GenDrop dc,
                                                              float inv_sqrt_d) {
    constexpr int d = 16;
    extern __shared__ __align__(16) float at_lds[];
    const int D = H * d, LD = D + 4;
    const int FP = (F + 15) & ~15, JT = FP >> 4;
    const int PL = FP * 4 + 4;                       // scratch plane (one per 4 queries of a tile): [key j][query & 3]
    float* sq = at_lds;
    float* sk = sq + FP * LD;
    float* sv = sk + FP * LD;
    float4* sst = reinterpret_cast<float4*>(sv + FP * LD);          // [H][FP]: max, 1/sum, dot, -
    float* scr = reinterpret_cast<float*>(sst + H * FP);            // [4 waves][4 PL]: dS^T, then Pm^T of the current query tile
    // (the go rows are read from global memory where they are needed - 8 scalar loads per lane and query tile, L1 / L2 hits
    // between the heads' waves - and the two transposed tiles share one scratch: 73 KB instead of 107 KB, two workgroups per CU)
    const int p = blockIdx.x;
    const int64_t base = (int64_t)p * F * D;
    for (int i = threadIdx.x; i < FP * (D >> 2); i += blockDim.x) {
        const int r = i / (D >> 2), c = (i - r * (D >> 2)) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, cc = a;
        if (r < F) {
            a = *reinterpret_cast<const float4*>(q + base + r * D + c);
            b = *reinterpret_cast<const float4*>(k + base + r * D + c);
            cc = *reinterpret_cast<const float4*>(v + base + r * D + c);
        }
        *reinterpret_cast<float4*>(sq + r * LD + c) = a;
        *reinterpret_cast<float4*>(sk + r * LD + c) = b;
        *reinterpret_cast<float4*>(sv + r * LD + c) = cc;
    }
    __syncthreads();
    for (int task = threadIdx.x; task < H * FP; task += blockDim.x) {
        const int h = task / FP, i = task - h * FP;
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);     // padding queries: P = exp2(0 - 0) * 0 = 0
        if (i < F) {
            f32x2 gi[d / 2], oi[d / 2];
            load_row<d>(go + base + (int64_t)i * D + h * d, gi);
            load_row<d>(o + base + (int64_t)i * D + h * d, oi);
            const float2 s2 = st[((size_t)p * H + h) * F + i];
            s4 = make_float4(s2.x, s2.y, dot_row<d>(gi, oi), 0.f);
        }
        sst[task] = s4;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    if (h >= H) return;
    const int b = order[p];
    const float sc_scale = kLog2e * inv_sqrt_d;
    const uint32_t skey = drop_sample_key(dc.key, (uint32_t)b);
    float* sds = scr + (size_t)h * 4 * PL;            // dS^T of the current query tile, then Pm^T
    const float* qh = sq + h * d;
    const float* kh = sk + h * d;
    const float* vh = sv + h * d;
    const float* gg_ = go + base + h * d;             // go rows of this head in global memory
    f32x4 dkt[4], dvt[4];                             // dK^T / dV^T[e = 4 g + r][key j = 16 jt + n]
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) { dkt[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int it = 0; it < JT; ++it) {
        const int i = 16 * it + n;
        const float4 s4 = sst[h * FP + i];
        float qb[4], gb[4];                           // second operands: Q^T / G^T[e = 4 ks + g][i]
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qb[ks] = qh[i * LD + 4 * ks + g]; gb[ks] = i < F ? gg_[(int64_t)i * D + 4 * ks + g] : 0.f; }
        f32x4 s[4], dp[4];                         // scores -> dS^T ; dP^T -> Pm^T
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            s[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (jt < JT) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    s[jt] = mfma4(kh[(16 * jt + n) * LD + 4 * ks + g], qb[ks], s[jt]);
                    dp[jt] = mfma4(vh[(16 * jt + n) * LD + 4 * ks + g], gb[ks], dp[jt]);
                }
            }
        }
        const uint32_t block0 = drop_attn_elem(h, F, i < F ? i : 0, 0) >> 2;
        f32x4 dqt = f32x4{0.f, 0.f, 0.f, 0.f};       // dQ^T[e = 4 g + r][i]
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            if (jt < JT) {
                const uint32_t kb = dc.on ? drop_keep4(skey, block0 + (uint32_t)(4 * jt + g), dc.thresh) : 0xFu;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * jt + 4 * g + r;
                    const float pr = j < F ? __builtin_amdgcn_exp2f(s[jt][r] * sc_scale - s4.x) * s4.y : 0.f;
                    const bool kp = (kb >> r) & 1u;
                    const float dpm = kp ? dp[jt][r] * dc.scale : 0.f;
                    const float ds = pr * (dpm - s4.z) * inv_sqrt_d;
                    const float pm = kp ? pr * dc.scale : 0.f;
                    s[jt][r] = ds;
                    dp[jt][r] = pm;
                    sds[(n >> 2) * PL + j * 4 + (n & 3)] = ds;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)           // contraction over keys j = 16 jt + 4 g + r: first operand K[j][e = n]
                    dqt = mfma4(kh[(16 * jt + 4 * g + r) * LD + n], s[jt][r], dqt);
            }
        }
        if (i < F)
            *reinterpret_cast<float4*>(dq + base + (int64_t)i * D + h * d + 4 * g) = make_float4(dqt[0], dqt[1], dqt[2], dqt[3]);
        __builtin_amdgcn_wave_barrier();
        // contraction over the tile's queries i = 16 it + 4 g + r: first operand Q / G[i][e = n], second dS^T / Pm^T[key 16 jt + n][i]
        float qa[4], ga[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ir = 16 * it + 4 * g + r;
            qa[r] = qh[ir * LD + n];
            ga[r] = ir < F ? gg_[(int64_t)ir * D + n] : 0.f;
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            if (jt < JT) {
                const float4 dsv = *reinterpret_cast<const float4*>(sds + g * PL + (16 * jt + n) * 4);
                dkt[jt] = mfma4(qa[0], dsv.x, dkt[jt]); dkt[jt] = mfma4(qa[1], dsv.y, dkt[jt]);
                dkt[jt] = mfma4(qa[2], dsv.z, dkt[jt]); dkt[jt] = mfma4(qa[3], dsv.w, dkt[jt]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the same scratch again for Pm^T
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
            if (jt < JT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sds[(n >> 2) * PL + (16 * jt + 4 * g + r) * 4 + (n & 3)] = dp[jt][r];
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            if (jt < JT) {
                const float4 pmv = *reinterpret_cast<const float4*>(sds + g * PL + (16 * jt + n) * 4);
                dvt[jt] = mfma4(ga[0], pmv.x, dvt[jt]); dvt[jt] = mfma4(ga[1], pmv.y, dvt[jt]);
                dvt[jt] = mfma4(ga[2], pmv.z, dvt[jt]); dvt[jt] = mfma4(ga[3], pmv.w, dvt[jt]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const int j = 16 * jt + n;
        if (jt < JT && j < F) {
            *reinterpret_cast<float4*>(dk + base + (int64_t)j * D + h * d + 4 * g) = make_float4(dkt[jt][0], dkt[jt][1], dkt[jt][2], dkt[jt][3]);
            *reinterpret_cast<float4*>(dv + base + (int64_t)j * D + h * d + 4 * g) = make_float4(dvt[jt][0], dvt[jt][1], dvt[jt][2], dvt[jt][3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
struct GenLayout {          // offsets in floats; nd = B F D, nu = B F U
    int64_t nd, nu;
    // saved by the forward, read by the backward
    int64_t xs, q0, k0, v, hq, hk, mq, mk, tq, tk, q, k, o, u, to, st, saved_total;
    // scratch
    int64_t dr, du, go, dq, dk, dv, dt, dm, dh, part, part_floats, ln_part, ln_blocks, modfull, mn_part, mn_slots, scratch_total;
};

static int gen_attn_mode() {      // 0 = automatic (MFMA forward where it is built), 1 = wavefront, 2 = MFMA
    static int mode = -1;
    if (mode < 0) mode = getenv("SATRANS_GENERIC_ATTN") ? atoi(getenv("SATRANS_GENERIC_ATTN")) : 0;
    return mode;
}
static int g_attn_override = -1;

static GenLayout gen_layout(const satrans_layer_desc* d) {
    GenLayout L;
    const int64_t N = (int64_t)d->B * d->F;
    L.nd = N * d->D;
    L.nu = N * std::max(d->U, 1);
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += (n + 3) & ~(int64_t)3; return r; };
    L.xs = take(L.nd); L.q0 = take(L.nd); L.k0 = take(L.nd); L.v = take(L.nd);
    L.hq = take(L.nu); L.hk = take(L.nu); L.mq = take(L.nd); L.mk = take(L.nd);
    L.tq = take(L.nd); L.tk = take(L.nd); L.q = take(L.nd); L.k = take(L.nd);
    L.o = take(L.nd); L.u = take(L.nd); L.to = take(L.nd); L.st = take(2 * (int64_t)d->B * d->H * d->F);
    L.saved_total = o;
    o = 0;
    L.dr = take(L.nd); L.du = take(L.nd); L.go = take(L.nd); L.dq = take(L.nd); L.dk = take(L.nd); L.dv = take(L.nd);
    L.dt = take(L.nd); L.dm = take(L.nd); L.dh = take(L.nu);
    const int64_t chunks = ceil_div(N, kTnRows) + d->S;
    // Partial sums of the weight-gradient products.  One region per product of a layer backward (their reductions run as ONE
    // launch at the end of the layer, gen_multi_reduce_kernel): dWo [D][D]; per role dW2 [U][D] and dW1 [D][U] per scenario; the
    // three [D][D] of dWq / dWk / dWv; the gamma / beta partials of up to three LayerNorm backward launches and of the two fused
    // MetaNet backward launches.  (gate / bilinear reduce behind each product and reuse the head of the region.)
    const int64_t Ue = std::max(d->U, d->D);
    L.mn_slots = ceil_div(N, kG2Rows) * std::max(d->S, 1);                 // gamma / beta partials of the fused MetaNet backward
    L.ln_blocks = std::min<int64_t>(1024, ceil_div(N, 256 / (d->D / 4)));
    L.part_floats = chunks * ((int64_t)d->D * d->D + 4 * (int64_t)d->S * d->D * Ue + 3 * (int64_t)d->D * d->D) +
                    3 * (L.ln_blocks * 2 * d->D + 4) + 2 * (L.mn_slots * 2 * d->D + 4) + 64;
    L.part = take(L.part_floats);
    L.ln_part = take(L.ln_blocks * 2 * d->D);
    L.modfull = take((d->flags & (SATRANS_GATE | SATRANS_BILINEAR)) ? (int64_t)d->S * d->D * d->D : 0);   // gate / bilinear: z^T g per scenario
    L.mn_part = take(L.mn_slots * 2 * d->D);
    L.scratch_total = o;
    return L;
}

static bool gen_supported(const satrans_layer_desc* d) {
    if (!d || ((d->flags & SATRANS_GATE) && (d->flags & SATRANS_BILINEAR))) return false;
    const int D = d->D, H = d->H;
    if (!(D == 16 || D == 32 || D == 64 || D == 128) || D % H) return false;
    const int dd = D / H;
    if (dd != 8 && dd != 16) return false;
    const bool meta = (d->flags & (SATRANS_META_Q | SATRANS_META_K)) && !(d->flags & (SATRANS_GATE | SATRANS_BILINEAR));
    if (meta && (d->U % 16 || d->U > 128 || (int64_t)D * d->U > 8192)) return false;
    if (D > 128 || d->F > 256) return false;
    if ((int64_t)4 * d->F * (D + 4) * 4 + (int64_t)H * d->F * 16 > 150 * 1024) return false;   // attention backward LDS
    return true;
}

static GenDrop gen_drop(const satrans_layer_desc* d, int site) {
    GenDrop g;
    g.on = (d->flags & SATRANS_TRAIN) && d->drop_p > 0.f;
    g.scale = g.on ? 1.0f / (1.0f - d->drop_p) : 1.0f;
    g.thresh = drop_threshold(d->drop_p);
    g.key = drop_site_key(d->seed, d->step, d->layer, site);
    return g;
}

template <bool TRANSB, int EPI>
static int gen_gemm(hipStream_t st, int batch, const float* const* A, const float* const* Bm, float* const* C, const int32_t* seg,
                    int S, int M, int F, int K, int N, int ldb, int64_t b_seg_stride, const float* mask = nullptr) {
    GemmBatch gb;
    for (int i = 0; i < 3; ++i) { gb.A[i] = A[i < batch ? i : 0]; gb.B[i] = Bm[i < batch ? i : 0]; gb.C[i] = C[i < batch ? i : 0]; }
    SATRANS_REQUIRE(K % 16 == 0 && N % 16 == 0 && K >= 16 && K <= 128 && N >= 16 && N <= 128, SATRANS_E_UNSUPPORTED,
                    "general-path product %d x %d: K and N must be multiples of 16 up to 128", K, N);
    const dim3 grid((unsigned)ceil_div(M, kG2Rows), (unsigned)(seg ? S : 1), (unsigned)batch);
    const size_t lds = sizeof(float) * 4 * (size_t)g2_plane_floats(K, N);
#define GEN_GEMM_CASE(KJ_)                                                                                                \
    case KJ_: {                                                                                                           \
        static bool attr_set = false;                                                                                     \
        if (!attr_set) {                                                                                                  \
            (void)hipFuncSetAttribute((const void*)gen_gemm_kernel<TRANSB, EPI, KJ_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                        \
            attr_set = true;                                                                                              \
        }                                                                                                                 \
        gen_gemm_kernel<TRANSB, EPI, KJ_><<<grid, 256, lds, st>>>(gb, seg, M, F, N, ldb, b_seg_stride, mask);              \
    } break;
    switch (K >> 4) {
        GEN_GEMM_CASE(1) GEN_GEMM_CASE(2) GEN_GEMM_CASE(3) GEN_GEMM_CASE(4) GEN_GEMM_CASE(5) GEN_GEMM_CASE(6) GEN_GEMM_CASE(7)
        GEN_GEMM_CASE(8)
    }
#undef GEN_GEMM_CASE
    SATRANS_CHECK_LAUNCH("gen_gemm_kernel");
    return SATRANS_OK;
}

// dst[s] += A^T G over segment s (seg == nullptr: one segment); dst_seg_stride between the segments' outputs
// `defer`: the product writes its partials into a region of its own and its reduction joins the layer's ONE reduction launch
// (gen_multi_reduce_kernel) instead of following the product
static int gen_gemm_tn(hipStream_t st, const float* A, const float* G, const int32_t* seg, int S, int M, int F, int K, int N,
                       float* partial, float* dst, int64_t dst_seg_stride, ReduceDefer* defer = nullptr) {
    const int segs = seg ? S : 1;
    const int chunks = (int)ceil_div(M, kTnRows);
    SATRANS_REQUIRE(K % 16 == 0 && N % 16 == 0 && K >= 16 && N >= 16 && K <= 128 && N <= 128, SATRANS_E_UNSUPPORTED,
                    "general-path weight gradient %d x %d: K and N must be multiples of 16 up to 128", K, N);
    const dim3 grid((unsigned)chunks, (unsigned)segs);
    // both above 64 (64 accumulator tiles): two passes over 64-row halves of dW, i.e. over column halves of A
    const int passes = K > 64 && N > 64 ? 2 : 1;
    for (int h = 0; h < passes; ++h) {
        const int Kp = passes == 1 ? K : (h == 0 ? 64 : K - 64);
        const float* Ap = A + 64 * h;
        float* dp = dst + (size_t)64 * h * N;
        const size_t lds = sizeof(float) * (size_t)Kp * N;
        if (defer) {
            partial = defer->take((int64_t)segs * chunks * Kp * N);
            SATRANS_REQUIRE(partial && defer->n < kMaxReduceJobs, SATRANS_E_LAUNCH, "general path: deferred reductions out of room");
        }
        if (Kp <= 64 && N <= 64) gen_gemm_tn_kernel<1, 1><<<grid, 256, lds, st>>>(Ap, G, seg, M, F, Kp, N, K, partial);
        else if (Kp <= 64) gen_gemm_tn_kernel<1, 2><<<grid, 256, lds, st>>>(Ap, G, seg, M, F, Kp, N, K, partial);
        else gen_gemm_tn_kernel<2, 1><<<grid, 256, lds, st>>>(Ap, G, seg, M, F, Kp, N, K, partial);
        SATRANS_CHECK_LAUNCH("gen_gemm_tn_kernel");
        if (defer) {
            defer->jobs.j[defer->n++] = ReduceJob{partial, dp, seg, dst_seg_stride, chunks, Kp * N, F, kTnRows, 0, segs, 0};
            continue;
        }
        gen_tn_reduce_kernel<<<dim3((unsigned)ceil_div((int64_t)Kp * N, 32), (unsigned)segs), 1024, 0, st>>>(partial, chunks, Kp * N, dp,
                                                                                                          dst_seg_stride, seg, F, kTnRows);
        SATRANS_CHECK_LAUNCH("gen_tn_reduce_kernel");
    }
    return SATRANS_OK;
}

// the collected reductions of a layer backward, one launch
// Jobs that add into the SAME destination (one generated-weight table for the Q and the K role: both roles' dW1 / dW2 and
// LayerNorm gradients accumulate into one buffer) must keep their order and may not run side by side: a job whose destination
// an earlier job of the launch already has goes to the next launch (two launches per layer then, the roles in their old order).
static int gen_run_deferred(hipStream_t st, ReduceDefer& defer) {
    bool done[kMaxReduceJobs] = {};
    int left = defer.n;
    while (left > 0) {
        ReduceJobs pass;
        int np = 0, bx = 1, by = 1;
        for (int i = 0; i < defer.n; ++i) {
            if (done[i]) continue;
            bool clash = false;
            for (int k = 0; k < np; ++k) clash = clash || pass.j[k].dst == defer.jobs.j[i].dst;
            // (a later job with the same destination as a POSTPONED one must wait as well: order is by first come)
            for (int k = 0; k < i; ++k) clash = clash || (!done[k] && defer.jobs.j[k].dst == defer.jobs.j[i].dst);
            if (clash) continue;
            pass.j[np++] = defer.jobs.j[i];
            bx = std::max(bx, (int)ceil_div(defer.jobs.j[i].count, 32));
            by = std::max(by, defer.jobs.j[i].kind == 0 ? defer.jobs.j[i].segs : 1);
        }
        for (int i = 0, k = 0; i < defer.n && k < np; ++i)
            if (!done[i] && pass.j[k].partial == defer.jobs.j[i].partial) { done[i] = true; ++k; }
        left -= np;
        gen_multi_reduce_kernel<<<dim3((unsigned)bx, (unsigned)by, (unsigned)np), 1024, 0, st>>>(pass);
        SATRANS_CHECK_LAUNCH("gen_multi_reduce_kernel");
    }
    defer.n = 0;
    return SATRANS_OK;
}

// dst_i += A^T G_i for three gradient tensors of width N = K = D <= 64 over the same rows A (see gen_gemm_tn_kernel, FAN)
static int gen_gemm_tn_fan(hipStream_t st, const float* A, const float* const* G, int M, int D, float* partial, float* const* dst,
                           ReduceDefer* defer = nullptr) {
    SATRANS_REQUIRE(D % 16 == 0 && D >= 16 && D <= 64, SATRANS_E_UNSUPPORTED, "general-path fan weight gradient: D = %d", D);
    const int chunks = (int)ceil_div(M, kTnRows);
    const size_t lds = sizeof(float) * 3 * (size_t)D * D;
    if (defer) {
        partial = defer->take((int64_t)chunks * 3 * D * D);
        SATRANS_REQUIRE(partial && defer->n + 3 <= kMaxReduceJobs, SATRANS_E_LAUNCH, "general path: deferred reductions out of room");
    }
    gen_gemm_tn_kernel<1, 3, true><<<dim3((unsigned)chunks, 1), 256, lds, st>>>(A, G[0], nullptr, M, 1, D, D, D, partial, G[1], G[2]);
    SATRANS_CHECK_LAUNCH("gen_gemm_tn_kernel(fan)");
    if (defer) {
        for (int w = 0; w < 3; ++w)
            defer->jobs.j[defer->n++] = ReduceJob{partial + (size_t)w * D * D, dst[w], nullptr, 0, chunks, D * D, 1, kTnRows, 3 * D * D, 1, 0};
        return SATRANS_OK;
    }
    for (int w = 0; w < 3; ++w) {
        gen_tn_reduce_kernel<<<dim3((unsigned)ceil_div((int64_t)D * D, 32), 1), 1024, 0, st>>>(partial + (size_t)w * D * D, chunks, D * D,
                                                                                             dst[w], 0, nullptr, 1, kTnRows, 3 * D * D);
        SATRANS_CHECK_LAUNCH("gen_tn_reduce_kernel");
    }
    return SATRANS_OK;
}

// C = (addend) + sum_i A_i W_i^T for nsrc = 3 or 4 sources with square [D, D] weights (see gen_gemm_sum_kernel)
static int gen_gemm_sum(hipStream_t st, int nsrc, const float* const* A, const float* const* W, int M, int D, const float* addend,
                        float* C) {
    SATRANS_REQUIRE((nsrc == 3 || nsrc == 4) && (D == 16 || D == 32 || D == 64), SATRANS_E_UNSUPPORTED,
                    "general-path summed product: %d sources, D = %d", nsrc, D);
    GemmBatch gb;
    for (int i = 0; i < 3; ++i) { gb.A[i] = A[i]; gb.B[i] = W[i]; gb.C[i] = C; }
    const dim3 grid((unsigned)ceil_div(M, kG2Rows));
    const size_t lds = sizeof(float) * 3 * 4 * (size_t)g2_plane_floats(D, D);
#define GEN_SUM_CASE(KJ_)                                                                                                      \
    case KJ_: {                                                                                                               \
        static bool attr_set = false;                                                                                         \
        if (!attr_set) {                                                                                                      \
            (void)hipFuncSetAttribute((const void*)gen_gemm_sum_kernel<KJ_, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                      160 * 1024);                                                                            \
            attr_set = true;                                                                                                  \
        }                                                                                                                     \
        gen_gemm_sum_kernel<KJ_, 3><<<grid, 256, lds, st>>>(gb, 3, M, addend, C);                                              \
    } break;
    switch (D >> 4) { GEN_SUM_CASE(1) GEN_SUM_CASE(2) GEN_SUM_CASE(4) }
#undef GEN_SUM_CASE
    SATRANS_CHECK_LAUNCH("gen_gemm_sum_kernel");
    if (nsrc == 4) {        // the fourth source (SelfAttention_Layer's residual projection) as one accumulating product
        const float* A4[1] = {A[3]};
        const float* B4[1] = {W[3]};
        float* C4[1] = {C};
        return gen_gemm<true, 2>(st, 1, A4, B4, C4, nullptr, 1, M, 1, D, D, D, 0);
    }
    return SATRANS_OK;
}

// the fused MetaNet forward where it is built: (D, U) in {(64, 128), (32, 64), (16, 32), (64, 16)}, planes of both weights <= 80 KB
static bool gen_metanet_fused_ok(int D, int U) {
    return (D == 64 && U == 128) || (D == 32 && U == 64) || (D == 16 && U == 32) || (D == 64 && U == 16);
}
static int gen_metanet_fused_fwd(hipStream_t st, const satrans_layer_desc* d, const float* z0, const float* tab, float* h, float* t,
                                 float* out, const float* gam, const float* bet, int site, bool norm) {
    const int M = d->B * d->F, D = d->D, U = d->U;
    const GenDrop dc = gen_drop(d, site);
    // 4 waves per workgroup (two workgroups = two waves per SIMD under the 78 KB of weight planes): 264 us per launch at the
    // configs[4] shape; 8 waves (four per SIMD) 290 us - more waves do not help, the kernel is bound by issue
    constexpr int wv = 4;
    const int rows = 128 * (wv == 8 ? 8 : 4);
    const dim3 grid((unsigned)ceil_div(M, rows), (unsigned)(d->seg ? d->S : 1));
    const size_t lds = sizeof(float) * 4 * ((size_t)g2_plane_floats(D, U) + (size_t)g2_plane_floats(U, D));
#define GEN_MN_CASE(KJ_, UJ_)                                                                                                   \
    {                                                                                                                            \
        static bool attr_set = false;                                                                                            \
        if (!attr_set) {                                                                                                         \
            (void)hipFuncSetAttribute((const void*)gen_metanet_fwd_kernel<KJ_, UJ_, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                               \
            (void)hipFuncSetAttribute((const void*)gen_metanet_fwd_kernel<KJ_, UJ_, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                               \
            attr_set = true;                                                                                                     \
        }                                                                                                                        \
        if (wv == 8)                                                                                                             \
            gen_metanet_fwd_kernel<KJ_, UJ_, 8><<<grid, 512, lds, st>>>(z0, tab, d->tab_stride, d->seg, M, d->F, h, t, out, gam, bet, \
                                                                      d->order, dc, norm);                                      \
        else                                                                                                                     \
            gen_metanet_fwd_kernel<KJ_, UJ_, 4><<<grid, 256, lds, st>>>(z0, tab, d->tab_stride, d->seg, M, d->F, h, t, out, gam, bet, \
                                                                      d->order, dc, norm);                                      \
    }
    if (D == 64 && U == 128) GEN_MN_CASE(4, 8)
    else if (D == 32 && U == 64) GEN_MN_CASE(2, 4)
    else if (D == 16 && U == 32) GEN_MN_CASE(1, 2)
    else GEN_MN_CASE(4, 1)
#undef GEN_MN_CASE
    SATRANS_CHECK_LAUNCH("gen_metanet_fwd_kernel");
    return SATRANS_OK;
}

// the fused output block (gen_out_ln_fwd_kernel): D in {16, 32, 64} (a token's row in the four lanes g of one wave, <= 16 registers);
// -DSATRANS_EXP_NO_OUT_LN: the product and the LayerNorm as two launches, as before
static bool gen_out_ln_fused_ok(int D) {
#ifdef SATRANS_EXP_NO_OUT_LN
    return false;
#else
    return D == 16 || D == 32 || D == 64;
#endif
}
static int gen_out_ln_fused_fwd(hipStream_t st, const satrans_layer_desc* d, const float* o, const float* w, const float* res, float* u,
                                float* t, float* y, bool y_orig, bool relu) {
    const int M = d->B * d->F, D = d->D;
    const GenDrop dc = gen_drop(d, kSiteOut);
    const unsigned grid = (unsigned)ceil_div(M, kG2Rows);
    const size_t lds = sizeof(float) * 4 * (size_t)g2_plane_floats(D, D);
#define GEN_OUT_CASE(KJ_)                                                                                                        \
    gen_out_ln_fwd_kernel<KJ_><<<grid, 256, lds, st>>>(o, w, res, u, t, y, y_orig, d->ln_g, d->ln_b, M, d->F, d->order, dc, relu);
    if (D == 64) { GEN_OUT_CASE(4) }
    else if (D == 32) { GEN_OUT_CASE(2) }
    else { GEN_OUT_CASE(1) }
#undef GEN_OUT_CASE
    SATRANS_CHECK_LAUNCH("gen_out_ln_fwd_kernel");
    return SATRANS_OK;
}

static int gen_metanet_fused_bwd(hipStream_t st, const satrans_layer_desc* d, float* g, const float* t, const float* h, const float* tab,
                                 float* dm, float* dh, const float* gam, float* part, int site) {
    const int M = d->B * d->F, D = d->D, U = d->U;
    const GenDrop dc = gen_drop(d, site);
    const dim3 grid((unsigned)ceil_div(M, kG2Rows), (unsigned)(d->seg ? d->S : 1));
    const size_t lds = sizeof(float) * std::max<size_t>(4 * ((size_t)g2_plane_floats(D, U) + (size_t)g2_plane_floats(U, D)), 2 * 64 * (size_t)D);
#define GEN_MNB_CASE(KJ_, UJ_)                                                                                                  \
    {                                                                                                                            \
        static bool attr_set = false;                                                                                            \
        if (!attr_set) {                                                                                                         \
            (void)hipFuncSetAttribute((const void*)gen_metanet_bwd_kernel<KJ_, UJ_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                               \
            attr_set = true;                                                                                                     \
        }                                                                                                                        \
        gen_metanet_bwd_kernel<KJ_, UJ_><<<grid, 256, lds, st>>>(g, t, h, tab, d->tab_stride, d->seg, M, d->F, dm, dh, gam, part,  \
                                                               d->order, dc);                                                   \
    }
    if (D == 64 && U == 128) GEN_MNB_CASE(4, 8)
    else if (D == 32 && U == 64) GEN_MNB_CASE(2, 4)
    else if (D == 16 && U == 32) GEN_MNB_CASE(1, 2)
    else GEN_MNB_CASE(4, 1)
#undef GEN_MNB_CASE
    SATRANS_CHECK_LAUNCH("gen_metanet_bwd_kernel");
    return SATRANS_OK;
}

#define GEN_LN_DISPATCH(D, CALL)                                   \
    switch ((D) / 4) {                                             \
        case 4: { constexpr int LPT = 4; CALL; } break;            \
        case 8: { constexpr int LPT = 8; CALL; } break;            \
        case 16: { constexpr int LPT = 16; CALL; } break;          \
        default: { constexpr int LPT = 32; CALL; } break;          \
    }

static int gen_ln_fwd(hipStream_t st, const satrans_layer_desc* d, const float* a, const float* res, float* t_out, float* y,
                      bool y_orig, const float* gamma, const float* beta, int site, bool relu, bool relu_post = false,
                      bool norm = true) {
    const int64_t N = (int64_t)d->B * d->F;
    const GenDrop dc = gen_drop(d, site);
    GEN_LN_DISPATCH(d->D, (gen_ln_fwd_kernel<LPT><<<(unsigned)ceil_div(N * LPT, 256), 256, 0, st>>>(
                              a, res, t_out, y, y_orig, gamma, beta, N, d->F, d->order, dc, relu, relu_post, norm)));
    SATRANS_CHECK_LAUNCH("gen_ln_fwd_kernel");
    return SATRANS_OK;
}

static int gen_ln_bwd(hipStream_t st, const satrans_layer_desc* d, const GenLayout& L, float* scratch, const float* g, bool g_orig,
                      const float* t, const float* a_pre, const float* gamma, float* dt, float* dm, int site, bool relu,
                      float* g_gamma_beta, bool relu_post = false, bool norm = true, ReduceDefer* defer = nullptr) {
    const int64_t N = (int64_t)d->B * d->F;
    const GenDrop dc = gen_drop(d, site);
    const int tpi = 256 / (d->D / 4);
    const int blocks = (int)L.ln_blocks;
    const int tpb = (int)(ceil_div(ceil_div(N, blocks), tpi) * tpi);
    float* part = scratch + L.ln_part;
    if (defer && g_gamma_beta && norm) {
        part = defer->take((int64_t)blocks * 2 * d->D);
        SATRANS_REQUIRE(part && defer->n < kMaxReduceJobs, SATRANS_E_LAUNCH, "general path: deferred reductions out of room");
    }
    GEN_LN_DISPATCH(d->D, (gen_ln_bwd_kernel<LPT><<<(unsigned)blocks, 256, 0, st>>>(g, g_orig, t, a_pre, gamma, dt, dm, part, N,
                                                                                   d->F, d->order, dc, relu, tpb, relu_post, norm)));
    SATRANS_CHECK_LAUNCH("gen_ln_bwd_kernel");
    if (g_gamma_beta && norm && defer) {
        defer->jobs.j[defer->n++] = ReduceJob{part, g_gamma_beta, nullptr, 0, blocks, 2 * d->D, 0, 1, 0, 1, 0};
    } else if (g_gamma_beta && norm) {
        gen_tn_reduce_kernel<<<dim3((unsigned)ceil_div(2 * d->D, 32), 1), 1024, 0, st>>>(part, blocks, 2 * d->D, g_gamma_beta, 0, nullptr, 0, 1);
        SATRANS_CHECK_LAUNCH("gen_tn_reduce_kernel");
    }
    return SATRANS_OK;
}

// attention forward / backward launches shared by the layer and by SelfAttention_Layer
static int gen_attention_fwd(hipStream_t st, const satrans_layer_desc* d, const float* q, const float* k, const float* v, float* o,
                             float2* stp, float* att, float inv_sqrt_d) {
    const int B = d->B, F = d->F, D = d->D, H = d->H;
    const GenDrop dc = gen_drop(d, kSiteAttn);
    const int dd = D / H;
    int mode = g_attn_override >= 0 ? g_attn_override : gen_attn_mode();
    const bool can_mfma = dd == 16 && F <= 64 && H <= 4 && !att;
    const bool mfma = mode == 2 ? can_mfma : (mode == 0 && can_mfma);
    SATRANS_REQUIRE(mode != 2 || can_mfma, SATRANS_E_UNSUPPORTED, "generic attention: the MFMA arm needs d = 16, F <= 64, H <= 4");
    if (mfma) {
        const int FP = (F + 15) & ~15;
        const size_t lds = sizeof(float) * 3 * FP * (D + 4);
        static size_t attr = 0;
        if (lds > attr) {
            hipError_t e = hipFuncSetAttribute((const void*)gen_attn_fwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "generic attention: LDS attribute: %s", hipGetErrorString(e));
            attr = lds;
        }
        gen_attn_fwd_mfma_kernel<<<B, 256, lds, st>>>(q, k, v, o, stp, F, H, d->order, dc, inv_sqrt_d);
        SATRANS_CHECK_LAUNCH("gen_attn_fwd_mfma_kernel");
    } else {
        const size_t lds = sizeof(float) * 2 * F * (D + 4);
        static size_t attr8 = 0, attr16 = 0;
        size_t& attr = dd == 8 ? attr8 : attr16;
        if (lds > attr) {
            hipError_t e = dd == 8 ? hipFuncSetAttribute((const void*)gen_attn_fwd_wave_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                   : hipFuncSetAttribute((const void*)gen_attn_fwd_wave_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "generic attention: LDS attribute: %s", hipGetErrorString(e));
            attr = lds;
        }
        if (dd == 8) gen_attn_fwd_wave_kernel<8><<<B, 256, lds, st>>>(q, k, v, o, stp, att, B, F, H, d->order, dc, inv_sqrt_d);
        else gen_attn_fwd_wave_kernel<16><<<B, 256, lds, st>>>(q, k, v, o, stp, att, B, F, H, d->order, dc, inv_sqrt_d);
        SATRANS_CHECK_LAUNCH("gen_attn_fwd_wave_kernel");
    }
    return SATRANS_OK;
}

static int gen_attention_bwd(hipStream_t st, const satrans_layer_desc* d, const float* q, const float* k, const float* v,
                             const float* o, const float* go, const float2* stp, float* dq, float* dk, float* dv, float inv_sqrt_d) {
    const int B = d->B, F = d->F, D = d->D, H = d->H;
    const GenDrop dc = gen_drop(d, kSiteAttn);
    const int dd = D / H;
    const int mode = g_attn_override >= 0 ? g_attn_override : gen_attn_mode();
    const bool can_mfma = dd == 16 && F <= 64 && H <= 4;
    SATRANS_REQUIRE(mode != 2 || can_mfma, SATRANS_E_UNSUPPORTED, "generic attention backward: the MFMA arm needs d = 16, F <= 64, H <= 4");
    if (can_mfma && mode != 1) {
        const int FP = (F + 15) & ~15;
        const size_t lds_m = sizeof(float) * (3 * (size_t)FP * (D + 4) + 4 * (size_t)H * FP + 4 * 4 * ((size_t)FP * 4 + 4));
        static size_t attr_m = 0;
        if (lds_m > attr_m) {
            hipError_t e = hipFuncSetAttribute((const void*)gen_attn_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
            SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "generic attention backward: LDS attribute: %s", hipGetErrorString(e));
            attr_m = lds_m;
        }
        gen_attn_bwd_mfma_kernel<<<B, 256, lds_m, st>>>(q, k, v, o, go, stp, dq, dk, dv, F, H, d->order, dc, inv_sqrt_d);
        SATRANS_CHECK_LAUNCH("gen_attn_bwd_mfma_kernel");
        return SATRANS_OK;
    }
    const size_t lds = sizeof(float) * (4 * (size_t)F * (D + 4) + 4 * (size_t)H * F);
    static size_t attr8 = 0, attr16 = 0;
    size_t& attr = dd == 8 ? attr8 : attr16;
    if (lds > attr) {
        hipError_t e = dd == 8 ? hipFuncSetAttribute((const void*)gen_attn_bwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                               : hipFuncSetAttribute((const void*)gen_attn_bwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "generic attention backward: LDS attribute: %s", hipGetErrorString(e));
        attr = lds;
    }
    if (dd == 8) gen_attn_bwd_kernel<8><<<B, 256, lds, st>>>(q, k, v, o, go, stp, dq, dk, dv, F, H, d->order, dc, inv_sqrt_d);
    else gen_attn_bwd_kernel<16><<<B, 256, lds, st>>>(q, k, v, o, go, stp, dq, dk, dv, F, H, d->order, dc, inv_sqrt_d);
    SATRANS_CHECK_LAUNCH("gen_attn_bwd_kernel");
    return SATRANS_OK;
}

}  // namespace satrans

using namespace satrans;

extern "C" int satrans_layer_generic_supported(const satrans_layer_desc* d) { return gen_supported(d) ? 1 : 0; }

// attention arm of the generic forward: 0 = automatic, 1 = one lane per query row ("wavefront"), 2 = MFMA (F <= 64, d = 16);
// -1 = back to the initial value (SATRANS_GENERIC_ATTN in the environment)
extern "C" int satrans_set_generic_attention(int mode) {
    SATRANS_REQUIRE(mode >= -1 && mode <= 2, SATRANS_E_BADARG, "set_generic_attention: %d", mode);
    satrans::g_attn_override = mode;
    return SATRANS_OK;
}

extern "C" int64_t satrans_layer_generic_saved_floats(const satrans_layer_desc* d) {
    return gen_supported(d) ? gen_layout(d).saved_total : -1;
}
extern "C" int64_t satrans_layer_generic_scratch_floats(const satrans_layer_desc* d) {
    return gen_supported(d) ? gen_layout(d).scratch_total : -1;
}

extern "C" int satrans_layer_fwd_generic(const satrans_layer_desc* d, float* y, float* att, float* saved, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SATRANS_REQUIRE(gen_supported(d), SATRANS_E_UNSUPPORTED, "layer_fwd(generic): shape not supported");
    SATRANS_REQUIRE(y && saved, SATRANS_E_BADARG, "layer_fwd(generic): null pointer");
    const GenLayout L = gen_layout(d);
    const int B = d->B, F = d->F, D = d->D, U = d->U, H = d->H, S = d->S, M = B * F;
    const bool meta_q = d->flags & SATRANS_META_Q, meta_k = d->flags & SATRANS_META_K;
    const bool relu = d->flags & SATRANS_RELU_OUT, use_res = !(d->flags & SATRANS_NO_RES);
    // SATRANS_X_SORTED: the input IS the scenario-sorted rows (the previous layer of a stack left them so: SATRANS_Y_SORTED
    // there) - no copy, the backward reads d->x again; else the rows are brought into that order (or gathered: layer 0)
    const bool x_sorted = d->flags & SATRANS_X_SORTED, y_sorted = d->flags & SATRANS_Y_SORTED;
    SATRANS_REQUIRE(!(x_sorted && d->x_rows), SATRANS_E_BADARG, "layer_fwd(generic): SATRANS_X_SORTED with a fused gather");
    float *q0 = saved + L.q0, *k0 = saved + L.k0, *v = saved + L.v;
    const float* xs = x_sorted ? d->x : saved + L.xs;
    int rc;
    if (!x_sorted) {
        gen_permute_in_kernel<<<(unsigned)ceil_div((int64_t)M * (D / 4), 256), 256, 0, st>>>(*d, nullptr, saved + L.xs, true);
        SATRANS_CHECK_LAUNCH("gen_permute_in_kernel");
    }
    {   // satrans.py:55-57
        const float* A[3] = {xs, xs, xs};
        const float* Bw[3] = {d->w_query, d->w_key, d->w_value};
        float* C[3] = {q0, k0, v};
                if ((rc = gen_gemm<false, 0>(st, 3, A, Bw, C, nullptr, 1, M, F, D, D, D, 0))) return rc;
    }
    constexpr bool fuse_metanet = true;
    auto metanet = [&](const float* z0, float* h, float* m, float* t, float* out, const float* tab, const float* gam,
                       const float* bet, int site) -> int {      // submodules.py:77-103
        if (fuse_metanet && gen_metanet_fused_ok(D, U))          // one pass: h and t saved for the backward, m never materialised
            return gen_metanet_fused_fwd(st, d, z0, tab, h, t, out, gam, bet, site, true);
        const float* A1[1] = {z0};
        const float* B1[1] = {tab};
        float* C1[1] = {h};
        int r = gen_gemm<false, 1>(st, 1, A1, B1, C1, d->seg, S, M, F, D, U, U, d->tab_stride);
        if (r) return r;
        const float* A2[1] = {h};
        const float* B2[1] = {tab + (size_t)D * U};
        float* C2[1] = {m};
        if ((r = gen_gemm<false, 0>(st, 1, A2, B2, C2, d->seg, S, M, F, U, D, D, d->tab_stride))) return r;
        return gen_ln_fwd(st, d, m, z0, t, out, false, gam, bet, site, false);
    };
    const float* q = q0;
    const float* k = k0;
    const bool gate = d->flags & SATRANS_GATE, bil = d->flags & SATRANS_BILINEAR;
    const unsigned mod_blocks = (unsigned)ceil_div((int64_t)M * D, 256);
    if (gate || bil) {      // satrans.py:61-64,68-69,79-81: the generated row applied directly (bilinear: queries only)
        if (bil) {
            gen_modulate_kernel<true, false><<<mod_blocks, 256, 0, st>>>(q0, d->tab_q, d->tab_stride, d->seg, S, F, D, D / H, saved + L.q, M);
            q = saved + L.q;
        } else {
            if (meta_q) {
                gen_modulate_kernel<false, false><<<mod_blocks, 256, 0, st>>>(q0, d->tab_q, d->tab_stride, d->seg, S, F, D, D / H, saved + L.q, M);
                q = saved + L.q;
            }
            if (meta_k) {
                gen_modulate_kernel<false, false><<<mod_blocks, 256, 0, st>>>(k0, d->tab_k, d->tab_stride, d->seg, S, F, D, D / H, saved + L.k, M);
                k = saved + L.k;
            }
        }
        SATRANS_CHECK_LAUNCH("gen_modulate_kernel");
    } else {
    if (meta_q) {
        if ((rc = metanet(q0, saved + L.hq, saved + L.mq, saved + L.tq, saved + L.q, d->tab_q, d->lnq_g, d->lnq_b, kSiteMetaQ))) return rc;
        q = saved + L.q;
    }
    if (meta_k) {
        if ((rc = metanet(k0, saved + L.hk, saved + L.mk, saved + L.tk, saved + L.k, d->tab_k, d->lnk_g, d->lnk_b, kSiteMetaK))) return rc;
        k = saved + L.k;
    }
    }
    if ((rc = gen_attention_fwd(st, d, q, k, v, saved + L.o, reinterpret_cast<float2*>(saved + L.st), att,   // satrans.py:75-90
                                1.0f / sqrtf((float)(D / H))))) return rc;
    if (gen_out_ln_fused_ok(D)) {   // satrans.py:91-99 in one launch (u only kept when the backward needs its sign: flag `relu`)
        return gen_out_ln_fused_fwd(st, d, saved + L.o, d->w_out, use_res ? xs : nullptr, relu ? saved + L.u : nullptr, saved + L.to, y,
                                    !y_sorted, relu);
    }
    {   // satrans.py:91-99
        const float* A[1] = {saved + L.o};
        const float* Bw[1] = {d->w_out};
        float* C[1] = {saved + L.u};
        if ((rc = gen_gemm<true, 0>(st, 1, A, Bw, C, nullptr, 1, M, F, D, D, D, 0))) return rc;
        if ((rc = gen_ln_fwd(st, d, saved + L.u, use_res ? xs : nullptr, saved + L.to, y, !y_sorted, d->ln_g, d->ln_b, kSiteOut, relu)))
            return rc;
    }
    return SATRANS_OK;
}

extern "C" int satrans_layer_bwd_generic(const satrans_layer_desc* d, const float* dy, float* dx, const float* saved,
                                         float* scratch, float* g_wq, float* g_wk, float* g_wv, float* g_wo, float* g_ln,
                                         float* g_lnq, float* g_lnk, float* g_tab_q, float* g_tab_k, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SATRANS_REQUIRE(gen_supported(d), SATRANS_E_UNSUPPORTED, "layer_bwd(generic): shape not supported");
    SATRANS_REQUIRE(dy && dx && saved && scratch && g_wq && g_wk && g_wv && g_wo && g_ln, SATRANS_E_BADARG, "layer_bwd(generic): null pointer");
    const GenLayout L = gen_layout(d);
    const int B = d->B, F = d->F, D = d->D, U = d->U, H = d->H, S = d->S, M = B * F;
    const bool meta_q = d->flags & SATRANS_META_Q, meta_k = d->flags & SATRANS_META_K;
    const bool relu = d->flags & SATRANS_RELU_OUT, use_res = !(d->flags & SATRANS_NO_RES);
    // (SATRANS_X_SORTED / SATRANS_Y_SORTED as in the forward: x, dx / dy in scenario-sorted order - nothing to permute)
    const bool x_sorted = d->flags & SATRANS_X_SORTED, y_sorted = d->flags & SATRANS_Y_SORTED;
    SATRANS_REQUIRE(!(x_sorted && d->x_rows), SATRANS_E_BADARG, "layer_bwd(generic): SATRANS_X_SORTED with a fused gather");
    const float *xs = x_sorted ? d->x : saved + L.xs, *q0 = saved + L.q0, *k0 = saved + L.k0, *v = saved + L.v;
    const bool gate = d->flags & SATRANS_GATE, bil = d->flags & SATRANS_BILINEAR;
    const bool q_mod = bil || meta_q, k_mod = !bil && meta_k;
    const float* q = q_mod ? saved + L.q : q0;
    const float* k = k_mod ? saved + L.k : k0;
    float *dr = x_sorted ? dx : scratch + L.dr, *du = scratch + L.du, *go = scratch + L.go, *dq = scratch + L.dq, *dk = scratch + L.dk,
          *dv = scratch + L.dv, *dt = scratch + L.dt, *dm = scratch + L.dm, *dh = scratch + L.dh, *part = scratch + L.part;
    int rc;
    // (the reductions of this layer's weight-gradient partials: collected, one launch at the end; gate / bilinear read one of
    //  their reduced products right away and keep the reduce-behind-the-product form)
    ReduceDefer defer_store;
    defer_store.part = part;
    defer_store.cap = L.part_floats;
    ReduceDefer* defer = (gate || bil) ? nullptr : &defer_store;
    // ---- output block: LayerNorm backward (dy arrives in the caller's sample order), dWo, go = du Wo ----------------------------
    // (the same fusion in this direction - LayerNorm backward + mask as the prologue of go = du Wo, one launch for two - was built
    //  and measured in round 6: 2.22 against 2.22-2.25 ms per layer backward, inside the run-to-run noise; not kept)
    if ((rc = gen_ln_bwd(st, d, L, scratch, dy, !y_sorted, saved + L.to, saved + L.u, d->ln_g, dr, du, kSiteOut, relu, g_ln, false, true,
                         defer))) return rc;
    if ((rc = gen_gemm_tn(st, du, saved + L.o, nullptr, 1, M, F, D, D, part, g_wo, 0, defer))) return rc;     // dWo[out][in] += du^T o
    {
        const float* A[1] = {du};
        const float* Bw[1] = {d->w_out};
        float* C[1] = {go};
        if ((rc = gen_gemm<false, 0>(st, 1, A, Bw, C, nullptr, 1, M, F, D, D, D, 0))) return rc;
    }
    if (!use_res) {
        hipError_t e = hipMemsetAsync(dr, 0, sizeof(float) * (size_t)L.nd, st);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "layer_bwd(generic): memset: %s", hipGetErrorString(e));
    }
    // ---- attention --------------------------------------------------------------------------------------------------------------
    if ((rc = gen_attention_bwd(st, d, q, k, v, saved + L.o, go, reinterpret_cast<const float2*>(saved + L.st), dq, dk, dv,
                                1.0f / sqrtf((float)(D / H))))) return rc;
    // ---- MetaNet backward of one role: g (gradient of the role's output rows) becomes the gradient of z0 -------------------------
    auto metanet_bwd = [&](float* g, const float* z0, const float* h, const float* t, const float* tab, const float* gam,
                           float* g_ln_role, float* g_tab, int site) -> int {
        constexpr bool fuse = true;
        if (fuse && gen_metanet_fused_ok(D, U)) {
            // data gradients in one pass (g <- dz0; dm, dh written for the two weight-gradient products), then dW2, dW1
            float* mn_part = defer ? defer->take(L.mn_slots * 2 * D) : scratch + L.mn_part;
            SATRANS_REQUIRE(mn_part && (!defer || defer->n < kMaxReduceJobs), SATRANS_E_LAUNCH, "general path: deferred reductions out of room");
            int r = gen_metanet_fused_bwd(st, d, g, t, h, tab, dm, dh, gam, mn_part, site);
            if (r) return r;
            if (g_ln_role && defer) {
                defer->jobs.j[defer->n++] = ReduceJob{mn_part, g_ln_role, d->seg, 0, (int)ceil_div(M, kG2Rows), 2 * D, F, kG2Rows, 0, S, 1};
            } else if (g_ln_role) {
                gen_slot_reduce_kernel<<<(unsigned)ceil_div(2 * D, 32), 1024, 0, st>>>(mn_part, (int)ceil_div(M, kG2Rows), S, d->seg, F,
                                                                                    kG2Rows, 2 * D, g_ln_role);
                SATRANS_CHECK_LAUNCH("gen_slot_reduce_kernel");
            }
            if ((r = gen_gemm_tn(st, h, dm, d->seg, S, M, F, U, D, part, g_tab + (size_t)D * U, d->tab_stride, defer))) return r;
            return gen_gemm_tn(st, z0, dh, d->seg, S, M, F, D, U, part, g_tab, d->tab_stride, defer);
        }
        int r = gen_ln_bwd(st, d, L, scratch, g, false, t, nullptr, gam, dt, dm, site, false, g_ln_role, false, true, defer);
        if (r) return r;
        // dW2[u][o] += h^T dm ;  dh = (dm W2^T) * [h > 0] ;  dW1[i][u] += z0^T dh ;  dz0 = dt + dh W1^T
        if ((r = gen_gemm_tn(st, h, dm, d->seg, S, M, F, U, D, part, g_tab + (size_t)D * U, d->tab_stride, defer))) return r;
        const float* A1[1] = {dm};
        const float* B1[1] = {tab + (size_t)D * U};
        float* C1[1] = {dh};
        if ((r = gen_gemm<true, 3>(st, 1, A1, B1, C1, d->seg, S, M, F, D, U, D, d->tab_stride, h))) return r;
        if ((r = gen_gemm_tn(st, z0, dh, d->seg, S, M, F, D, U, part, g_tab, d->tab_stride, defer))) return r;
        const float* A2[1] = {dh};
        const float* B2[1] = {tab};
        float* C2[1] = {g};                  // g <- dt + dh W1^T
        return gen_gemm<true, 2>(st, 1, A2, B2, C2, d->seg, S, M, F, U, D, U, d->tab_stride, dt);
    };
    // ---- gate / bilinear backward of one role: d row from the segment's z0^T g (diagonal / diagonal blocks), g <- gradient of z0 -----
    auto modulate_bwd = [&](float* g, const float* z0, const float* tab, float* g_tab) -> int {
        SATRANS_REQUIRE(g_tab, SATRANS_E_BADARG, "layer_bwd(generic): null gradient of the generated rows");
        float* full = scratch + L.modfull;
        hipError_t e = hipMemsetAsync(full, 0, sizeof(float) * (size_t)S * D * D, st);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "layer_bwd(generic): memset: %s", hipGetErrorString(e));
        int r = gen_gemm_tn(st, z0, g, d->seg, S, M, F, D, D, part, full, (int64_t)D * D);
        if (r) return r;
        const int len = bil ? D * (D / H) : D;
        const unsigned eb = (unsigned)ceil_div((int64_t)S * len, 256), mb = (unsigned)ceil_div((int64_t)M * D, 256);
        if (bil) {
            gen_modulate_extract_kernel<true><<<eb, 256, 0, st>>>(full, S, D, D / H, g_tab, d->tab_stride);
            gen_modulate_kernel<true, true><<<mb, 256, 0, st>>>(g, tab, d->tab_stride, d->seg, S, F, D, D / H, dt, M);
        } else {
            gen_modulate_extract_kernel<false><<<eb, 256, 0, st>>>(full, S, D, D / H, g_tab, d->tab_stride);
            gen_modulate_kernel<false, true><<<mb, 256, 0, st>>>(g, tab, d->tab_stride, d->seg, S, F, D, D / H, dt, M);
        }
        SATRANS_CHECK_LAUNCH("gen_modulate_kernel(backward)");
        e = hipMemcpyAsync(g, dt, sizeof(float) * (size_t)L.nd, hipMemcpyDeviceToDevice, st);
        SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "layer_bwd(generic): copy: %s", hipGetErrorString(e));
        return SATRANS_OK;
    };
    if (gate || bil) {
        if (q_mod && (rc = modulate_bwd(dq, q0, d->tab_q, g_tab_q))) return rc;
        if (k_mod && (rc = modulate_bwd(dk, k0, d->tab_k, g_tab_k))) return rc;
    } else {
    if (meta_q && (rc = metanet_bwd(dq, q0, saved + L.hq, saved + L.tq, d->tab_q, d->lnq_g, g_lnq, g_tab_q, kSiteMetaQ))) return rc;
    if (meta_k && (rc = metanet_bwd(dk, k0, saved + L.hk, saved + L.tk, d->tab_k, d->lnk_g, g_lnk, g_tab_k, kSiteMetaK))) return rc;
    }
    // ---- projections: dW{q,k,v}[i][o] += x^T g ;  dx = dr + gq Wq^T + gk Wk^T + gv Wv^T ---------------------------------------------
    if (D <= 64) {
        const float* gs[3] = {dq, dk, dv};
        float* dsts[3] = {g_wq, g_wk, g_wv};
        if ((rc = gen_gemm_tn_fan(st, xs, gs, M, D, part, dsts, defer))) return rc;
    } else {
        if ((rc = gen_gemm_tn(st, xs, dq, nullptr, 1, M, F, D, D, part, g_wq, 0, defer))) return rc;
        if ((rc = gen_gemm_tn(st, xs, dk, nullptr, 1, M, F, D, D, part, g_wk, 0, defer))) return rc;
        if ((rc = gen_gemm_tn(st, xs, dv, nullptr, 1, M, F, D, D, part, g_wv, 0, defer))) return rc;
    }
    if (defer && (rc = gen_run_deferred(st, *defer))) return rc;
    {
        const float* grads[3] = {dq, dk, dv};
        const float* ws[3] = {d->w_query, d->w_key, d->w_value};
        if (D <= 64) {
            if ((rc = gen_gemm_sum(st, 3, grads, ws, M, D, dr, dr))) return rc;
        } else {
            for (int i = 0; i < 3; ++i) {
                const float* A[1] = {grads[i]};
                const float* Bw[1] = {ws[i]};
                float* C[1] = {dr};
                if ((rc = gen_gemm<true, 2>(st, 1, A, Bw, C, nullptr, 1, M, F, D, D, D, 0))) return rc;
            }
        }
    }
    if (!x_sorted) {      // (sorted input: the sums above went straight into dx)
        gen_permute_out_kernel<<<(unsigned)ceil_div((int64_t)M * (D / 4), 256), 256, 0, st>>>(*d, dr, dx);
        SATRANS_CHECK_LAUNCH("gen_permute_out_kernel");
    }
    return SATRANS_OK;
}

// =================================================================================================================================
// Sibling users of the same kernels
//   * SelfAttention_Layer (reference models/submodules.py:178-238; `usetrans` of star.py, mmoe.py, ple.py, sharedbottom.py,
//     adasparse.py): q,k,v = x W ; multi-head attention ; y = LayerNorm(relu(drop(o) + x W_Res)) - no MetaNet, no out-linear
//     (W_Out is a parameter the reference never uses), ReLU BEFORE the norm, a projected residual.
//   * MetaNet over an embedding block = BaseModel.meta_transformation (models/basemodel.py:191-199; `metatrans` of deepfm.py,
//     dcn.py, ...): y = [LayerNorm](drop(relu(x W1[s]) W2[s]) + x) with the scenario's generated weights.
// =================================================================================================================================
namespace satrans {

__global__ void gen_iota_kernel(int32_t* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

struct SelfAttLayout {
    int64_t nd, order, q, k, v, o, r, t, st, saved_total, dt, dm, dq, dk, dv, part, ln_part, scratch_total;
    int ln_blocks;
};
static SelfAttLayout selfatt_layout(const satrans_selfatt_desc* d) {
    SelfAttLayout L;
    const int64_t N = (int64_t)d->B * d->F;
    L.nd = N * d->D;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += (n + 3) & ~(int64_t)3; return r; };
    L.order = take(d->B); L.q = take(L.nd); L.k = take(L.nd); L.v = take(L.nd); L.o = take(L.nd); L.r = take(L.nd); L.t = take(L.nd);
    L.st = take(2 * (int64_t)d->B * d->H * d->F);
    L.saved_total = o;
    o = 0;
    L.dt = take(L.nd); L.dm = take(L.nd); L.dq = take(L.nd); L.dk = take(L.nd); L.dv = take(L.nd);
    L.part = take((ceil_div(N, kTnRows) + 1) * 3 * (int64_t)d->D * d->D);      // (three [D][D] partials per chunk: gen_gemm_tn_fan)
    L.ln_blocks = (int)std::min<int64_t>(1024, ceil_div(N, 256 / (d->D / 4)));
    L.ln_part = take((int64_t)L.ln_blocks * 2 * d->D);
    L.scratch_total = o;
    return L;
}
static bool selfatt_supported(const satrans_selfatt_desc* d) {
    if (!d || d->B <= 0 || d->F <= 0 || d->H <= 0) return false;
    if (!(d->D == 16 || d->D == 32 || d->D == 64 || d->D == 128) || d->D % d->H) return false;
    const int dd = d->D / d->H;
    return (dd == 8 || dd == 16) && (int64_t)4 * d->F * (d->D + 4) * 4 + (int64_t)d->H * d->F * 16 <= 150 * 1024;
}
// the pieces of a layer descriptor the shared helpers read
static satrans_layer_desc as_layer(const satrans_selfatt_desc* d, const int32_t* order) {
    satrans_layer_desc l;
    memset(&l, 0, sizeof(l));
    l.B = d->B; l.F = d->F; l.D = d->D; l.H = d->H; l.S = 1;
    l.flags = d->flags & SATRANS_TRAIN;
    l.drop_p = d->drop_p; l.seed = d->seed; l.step = d->step; l.layer = d->layer;
    l.order = order;
    return l;
}

}  // namespace satrans

extern "C" int64_t satrans_selfatt_saved_floats(const satrans_selfatt_desc* d) {
    return selfatt_supported(d) ? selfatt_layout(d).saved_total : -1;
}
extern "C" int64_t satrans_selfatt_scratch_floats(const satrans_selfatt_desc* d) {
    return selfatt_supported(d) ? selfatt_layout(d).scratch_total : -1;
}

extern "C" int satrans_selfatt_fwd(const satrans_selfatt_desc* d, float* y, float* att, float* saved, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SATRANS_REQUIRE(selfatt_supported(d), SATRANS_E_UNSUPPORTED, "selfatt_fwd: shape not supported");
    const bool use_res = !(d->flags & SATRANS_NO_RES);
    SATRANS_REQUIRE(d->x && d->w_query && d->w_key && d->w_value && d->ln_g && d->ln_b && y && saved && (!use_res || d->w_res),
                    SATRANS_E_BADARG, "selfatt_fwd: null pointer");
    const SelfAttLayout L = selfatt_layout(d);
    const int B = d->B, F = d->F, D = d->D, M = B * F;
    int32_t* order = reinterpret_cast<int32_t*>(saved + L.order);
    gen_iota_kernel<<<(unsigned)ceil_div(B, 256), 256, 0, st>>>(order, B);
    SATRANS_CHECK_LAUNCH("gen_iota_kernel");
    const satrans_layer_desc ld = as_layer(d, order);
    int rc;
    {
        const float* A[3] = {d->x, d->x, d->x};
        const float* Bw[3] = {d->w_query, d->w_key, d->w_value};
        float* C[3] = {saved + L.q, saved + L.k, saved + L.v};
                if ((rc = gen_gemm<false, 0>(st, 3, A, Bw, C, nullptr, 1, M, F, D, D, D, 0))) return rc;
    }
    const float scale = (d->flags & SATRANS_NO_SCALING) ? 1.0f : 1.0f / sqrtf((float)(D / d->H));
    if ((rc = gen_attention_fwd(st, &ld, saved + L.q, saved + L.k, saved + L.v, saved + L.o, reinterpret_cast<float2*>(saved + L.st),
                                att, scale))) return rc;
    if (use_res) {
        const float* A[1] = {d->x};
        const float* Bw[1] = {d->w_res};
        float* C[1] = {saved + L.r};
        if ((rc = gen_gemm<false, 0>(st, 1, A, Bw, C, nullptr, 1, M, F, D, D, D, 0))) return rc;
    }
    return gen_ln_fwd(st, &ld, saved + L.o, use_res ? saved + L.r : nullptr, saved + L.t, y, false, d->ln_g, d->ln_b, kSiteOut, false,
                      true, true);
}

extern "C" int satrans_selfatt_bwd(const satrans_selfatt_desc* d, const float* dy, float* dx, const float* saved, float* scratch,
                                   float* g_wq, float* g_wk, float* g_wv, float* g_wres, float* g_ln, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SATRANS_REQUIRE(selfatt_supported(d), SATRANS_E_UNSUPPORTED, "selfatt_bwd: shape not supported");
    const bool use_res = !(d->flags & SATRANS_NO_RES);
    SATRANS_REQUIRE(dy && dx && saved && scratch && g_wq && g_wk && g_wv && g_ln && (!use_res || g_wres), SATRANS_E_BADARG,
                    "selfatt_bwd: null pointer");
    const SelfAttLayout L = selfatt_layout(d);
    const int B = d->B, F = d->F, D = d->D, M = B * F;
    const int32_t* order = reinterpret_cast<const int32_t*>(saved + L.order);
    const satrans_layer_desc ld = as_layer(d, order);
    GenLayout GL;
    memset(&GL, 0, sizeof(GL));
    GL.ln_blocks = L.ln_blocks;
    GL.ln_part = L.ln_part;
    float *dt = scratch + L.dt, *dm = scratch + L.dm, *dq = scratch + L.dq, *dk = scratch + L.dk, *dv = scratch + L.dv,
          *part = scratch + L.part;
    int rc;
    // y = LN(relu(t)), t = drop(o) + x W_Res: dt = gradient of t (zero where t <= 0), dm = dt * dropout mask = gradient of o
    if ((rc = gen_ln_bwd(st, &ld, GL, scratch, dy, false, saved + L.t, nullptr, d->ln_g, dt, dm, kSiteOut, false, g_ln, true, true)))
        return rc;
    const float scale = (d->flags & SATRANS_NO_SCALING) ? 1.0f : 1.0f / sqrtf((float)(D / d->H));
    if ((rc = gen_attention_bwd(st, &ld, saved + L.q, saved + L.k, saved + L.v, saved + L.o, dm,
                                reinterpret_cast<const float2*>(saved + L.st), dq, dk, dv, scale))) return rc;
    if (D <= 64) {
        const float* gs[3] = {dq, dk, dv};
        float* dsts[3] = {g_wq, g_wk, g_wv};
        if ((rc = gen_gemm_tn_fan(st, d->x, gs, M, D, part, dsts))) return rc;
    } else {
        if ((rc = gen_gemm_tn(st, d->x, dq, nullptr, 1, M, F, D, D, part, g_wq, 0))) return rc;
        if ((rc = gen_gemm_tn(st, d->x, dk, nullptr, 1, M, F, D, D, part, g_wk, 0))) return rc;
        if ((rc = gen_gemm_tn(st, d->x, dv, nullptr, 1, M, F, D, D, part, g_wv, 0))) return rc;
    }
    if (use_res && (rc = gen_gemm_tn(st, d->x, dt, nullptr, 1, M, F, D, D, part, g_wres, 0))) return rc;
    // dx = dq Wq^T + dk Wk^T + dv Wv^T (+ dt Wres^T)
    const float* grads[4] = {dq, dk, dv, dt};
    const float* ws[4] = {d->w_query, d->w_key, d->w_value, d->w_res};
    if (D <= 64) return gen_gemm_sum(st, use_res ? 4 : 3, grads, ws, M, D, nullptr, dx);
    for (int i = 0; i < (use_res ? 4 : 3); ++i) {
        const float* A[1] = {grads[i]};
        const float* Bw[1] = {ws[i]};
        float* C[1] = {dx};
        rc = i == 0 ? gen_gemm<true, 0>(st, 1, A, Bw, C, nullptr, 1, M, F, D, D, D, 0)
                    : gen_gemm<true, 2>(st, 1, A, Bw, C, nullptr, 1, M, F, D, D, D, 0);
        if (rc) return rc;
    }
    return SATRANS_OK;
}

// ---- MetaNet over an embedding block ----------------------------------------------------------------------------------------------
namespace satrans {
struct MetaLayout {
    int64_t nd, nu, xs, h, m, t, saved_total, dt, dm, dh, part, ln_part, scratch_total;
    int ln_blocks;
};
static MetaLayout metanet_layout(const satrans_metanet_desc* d) {
    MetaLayout L;
    const int64_t N = (int64_t)d->B * d->F;
    L.nd = N * d->D;
    L.nu = N * d->U;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += (n + 3) & ~(int64_t)3; return r; };
    L.xs = take(L.nd); L.h = take(L.nu); L.m = take(L.nd); L.t = take(L.nd);
    L.saved_total = o;
    o = 0;
    L.dt = take(L.nd); L.dm = take(L.nd); L.dh = take(L.nu);
    L.part = take((int64_t)d->S * (ceil_div(N, kTnRows) + 1) * (int64_t)d->D * d->U);
    L.ln_blocks = (int)std::min<int64_t>(1024, ceil_div(N, 256 / (d->D / 4)));
    L.ln_part = take((int64_t)L.ln_blocks * 2 * d->D);
    L.scratch_total = o;
    return L;
}
static bool metanet_supported(const satrans_metanet_desc* d) {
    if (!d || d->B <= 0 || d->F <= 0 || d->S <= 0) return false;
    if (!(d->D == 16 || d->D == 32 || d->D == 64 || d->D == 128)) return false;
    return d->U > 0 && d->U % 16 == 0 && d->U <= 128 && (int64_t)d->D * d->U <= 8192;
}
static satrans_layer_desc as_layer(const satrans_metanet_desc* d) {
    satrans_layer_desc l;
    memset(&l, 0, sizeof(l));
    l.B = d->B; l.F = d->F; l.D = d->D; l.H = 1; l.U = d->U; l.S = d->S;
    l.flags = d->flags & SATRANS_TRAIN;
    l.drop_p = d->drop_p; l.seed = d->seed; l.step = d->step; l.layer = d->layer;
    l.order = d->order; l.seg = d->seg; l.x = d->x; l.tab_stride = d->tab_stride;
    return l;
}
}  // namespace satrans

extern "C" int64_t satrans_metanet_saved_floats(const satrans_metanet_desc* d) {
    return metanet_supported(d) ? metanet_layout(d).saved_total : -1;
}
extern "C" int64_t satrans_metanet_scratch_floats(const satrans_metanet_desc* d) {
    return metanet_supported(d) ? metanet_layout(d).scratch_total : -1;
}

extern "C" int satrans_metanet_fwd(const satrans_metanet_desc* d, float* y, float* saved, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SATRANS_REQUIRE(metanet_supported(d), SATRANS_E_UNSUPPORTED, "metanet_fwd: shape not supported");
    const bool norm = !(d->flags & SATRANS_NO_NORM);
    SATRANS_REQUIRE(d->x && d->order && d->seg && d->tab && y && saved && (!norm || (d->ln_g && d->ln_b)), SATRANS_E_BADARG,
                    "metanet_fwd: null pointer");
    const MetaLayout L = metanet_layout(d);
    const satrans_layer_desc ld = as_layer(d);
    const int M = d->B * d->F, D = d->D, U = d->U;
    gen_permute_in_kernel<<<(unsigned)ceil_div((int64_t)M * (D / 4), 256), 256, 0, st>>>(ld, nullptr, saved + L.xs, true);
    SATRANS_CHECK_LAUNCH("gen_permute_in_kernel");
    int rc;
    const float* A1[1] = {saved + L.xs};
    const float* B1[1] = {d->tab};
    float* C1[1] = {saved + L.h};
    if ((rc = gen_gemm<false, 1>(st, 1, A1, B1, C1, d->seg, d->S, M, d->F, D, U, U, d->tab_stride))) return rc;
    const float* A2[1] = {saved + L.h};
    const float* B2[1] = {d->tab + (size_t)D * U};
    float* C2[1] = {saved + L.m};
    if ((rc = gen_gemm<false, 0>(st, 1, A2, B2, C2, d->seg, d->S, M, d->F, U, D, D, d->tab_stride))) return rc;
    return gen_ln_fwd(st, &ld, saved + L.m, saved + L.xs, saved + L.t, y, true, d->ln_g, d->ln_b, kSiteMetaQ, false, false, norm);
}

extern "C" int satrans_metanet_bwd(const satrans_metanet_desc* d, const float* dy, float* dx, const float* saved, float* scratch,
                                   float* g_tab, float* g_ln, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SATRANS_REQUIRE(metanet_supported(d), SATRANS_E_UNSUPPORTED, "metanet_bwd: shape not supported");
    const bool norm = !(d->flags & SATRANS_NO_NORM);
    SATRANS_REQUIRE(dy && dx && saved && scratch && g_tab && (!norm || g_ln), SATRANS_E_BADARG, "metanet_bwd: null pointer");
    const MetaLayout L = metanet_layout(d);
    const satrans_layer_desc ld = as_layer(d);
    const int M = d->B * d->F, D = d->D, U = d->U, S = d->S, F = d->F;
    GenLayout GL;
    memset(&GL, 0, sizeof(GL));
    GL.ln_blocks = L.ln_blocks;
    GL.ln_part = L.ln_part;
    float *dt = scratch + L.dt, *dm = scratch + L.dm, *dh = scratch + L.dh, *part = scratch + L.part;
    int rc;
    if ((rc = gen_ln_bwd(st, &ld, GL, scratch, dy, true, saved + L.t, nullptr, d->ln_g, dt, dm, kSiteMetaQ, false, g_ln, false, norm)))
        return rc;
    if ((rc = gen_gemm_tn(st, saved + L.h, dm, d->seg, S, M, F, U, D, part, g_tab + (size_t)D * U, d->tab_stride))) return rc;
    const float* A1[1] = {dm};
    const float* B1[1] = {d->tab + (size_t)D * U};
    float* C1[1] = {dh};
    if ((rc = gen_gemm<true, 3>(st, 1, A1, B1, C1, d->seg, S, M, F, D, U, D, d->tab_stride, saved + L.h))) return rc;
    if ((rc = gen_gemm_tn(st, saved + L.xs, dh, d->seg, S, M, F, D, U, part, g_tab, d->tab_stride))) return rc;
    const float* A2[1] = {dh};
    const float* B2[1] = {d->tab};
    float* C2[1] = {dt};
    if ((rc = gen_gemm<true, 2>(st, 1, A2, B2, C2, d->seg, S, M, F, U, D, U, d->tab_stride))) return rc;
    gen_permute_out_kernel<<<(unsigned)ceil_div((int64_t)M * (D / 4), 256), 256, 0, st>>>(ld, dt, dx);
    SATRANS_CHECK_LAUNCH("gen_permute_out_kernel");
    return SATRANS_OK;
}
