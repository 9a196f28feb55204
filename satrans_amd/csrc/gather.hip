// K0 scenario bucketing + K1 fused multi-table embedding gather.
//
// K1 replaces the reference's 19 separate nn.Embedding launches and the torch.cat that follows
// (models/meta_basemodel.py:533-535, models/satrans.py:211) by ONE launch that writes [B,F,D] directly.
// HBM-bound: every gathered row is a random D*4-byte read (128 B at D=32).  A row is moved by D/4 lanes
// with one 16-byte load each, so one wave instruction moves 64/(D/4) rows (8 rows at D=32) and each
// 128-byte row is one fully used cache line.  Every thread keeps kRowsPerThread independent rows in
// flight to cover the ~900-cycle HBM miss latency (MI355X_MICROARCH.md, cycle constants).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace satrans {

constexpr int kGatherBlock = 256;
constexpr int kRowsPerThread = 4;

template <int LPR>  // lanes per row = D/4
__global__ __launch_bounds__(kGatherBlock) void gather_rows_kernel(
    const float4* __restrict__ arena, const int64_t* __restrict__ row_span, const int32_t* __restrict__ cols,
    const void* __restrict__ X, int id_dtype, int64_t x_stride, int64_t n_rows, int F, float4* __restrict__ out,
    int32_t* __restrict__ rows_out, int32_t* __restrict__ status) {
    const int64_t tid = (int64_t)blockIdx.x * kGatherBlock + threadIdx.x;
    const int64_t slot0 = tid / LPR;        // first (sample, field) pair of this thread
    const int q = (int)(tid % LPR);         // which 16-byte piece of the row
    const int64_t stride = (int64_t)gridDim.x * kGatherBlock / LPR;

    // (sample, field) of a slot without a division per slot: one division for the first slot and one for the stride, then
    // incremental updates (a runtime integer division costs as much as the rest of a slot's address arithmetic)
    // (32-bit: n_rows = B * F < 2^31 is checked by the caller; a 64-bit division would cost several times more still)
    const int sb = (int)stride / F;
    const int sf = (int)stride - sb * F;
    int b0 = (int)min(slot0, n_rows) / F;
    int f0 = (int)min(slot0, n_rows) - b0 * F;
    for (int64_t base = slot0; base < n_rows; base += stride * kRowsPerThread) {
        int64_t row[kRowsPerThread];
        bool bad[kRowsPerThread];
        float4 val[kRowsPerThread];
#pragma unroll
        for (int r = 0; r < kRowsPerThread; ++r) {
            const int64_t slot = base + r * stride;
            row[r] = -1;
            bad[r] = false;
            if (slot < n_rows) {
                const int64_t b = b0;
                const int f = f0;
                const int64_t id = load_id(X, id_dtype, x_stride, b, cols[f]);
                const int64_t lo = row_span[2 * f], hi = row_span[2 * f + 1];
                // out of range: the reference raises IndexError; we flag, write zeros and record the table's first row
                // (the recorded row stays inside the field's own table, which the optimizer's table classes rely on)
                bad[r] = id < 0 || id >= hi - lo;
                row[r] = bad[r] ? lo : lo + id;
            }
            b0 += sb;                       // next slot of this thread: + stride
            f0 += sf;
            if (f0 >= F) { f0 -= F; ++b0; }
        }
#pragma unroll
        for (int r = 0; r < kRowsPerThread; ++r) {
            val[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row[r] >= 0 && !bad[r] && out) val[r] = arena[row[r] * LPR + q];
        }
#pragma unroll
        for (int r = 0; r < kRowsPerThread; ++r) {
            const int64_t slot = base + r * stride;
            if (slot < n_rows) {
                if (out) out[slot * LPR + q] = val[r];
                if (q == 0) {
                    if (rows_out) rows_out[slot] = (int32_t)row[r];
                    if (bad[r]) atomicOr(status, 1);
                }
            }
        }
    }
}

__global__ void extract_scenario_kernel(const void* __restrict__ X, int id_dtype, int64_t x_stride, int col, int B,
                                        int S, int32_t* __restrict__ sid, uint32_t* __restrict__ keys,
                                        int32_t* __restrict__ idx, int32_t* __restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int64_t s = load_id(X, id_dtype, x_stride, b, col);
    if (s < 0 || s >= S) {
        atomicOr(status, 1);
        s = 0;
    }
    sid[b] = (int32_t)s;
    keys[b] = (uint32_t)s;
    idx[b] = b;
}

// seg[s] = first position in the sorted key array whose key is >= s;  seg[S] = B
__global__ void segment_bounds_kernel(const uint32_t* __restrict__ sorted_keys, int B, int S,
                                      int32_t* __restrict__ seg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > B) return;
    const int64_t prev = (i == 0) ? -1 : (int64_t)sorted_keys[i - 1];
    const int64_t cur = (i == B) ? (int64_t)S : (int64_t)sorted_keys[i];
    for (int64_t s = prev + 1; s <= cur; ++s) seg[s] = i;
}

// The whole bucketing as ONE launch for a handful of scenarios (the usual case: AliCCP has 4 scenario rows): a stable counting sort
// in one workgroup.  Samples are taken in rounds of 1,024 (thread t of round r owns sample 1024 r + t: coalesced reads); inside a
// wave the samples of scenario s are ranked by a ballot, so the counts c[s][r][w] (scenario, round, wave - in that order) ARE the
// histogram of a stable sort, and their exclusive prefix plus the rank inside the wave is a sample's position in `order`.
// (The general path below is extract + a rocPRIM radix sort of six launches + bounds: eight launches of ~5 us.)
constexpr int kBucketThreads = 1024, kBucketWaves = kBucketThreads / 64, kBucketMaxS = 16, kBucketMaxRounds = 64;
__device__ __forceinline__ int lanes_below(uint64_t mask) {      // set bits of `mask` in lanes below this one
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
template <int SP>      // scenario rows, padded to 4 / 8 / 16: the ballots per round
__global__ __launch_bounds__(kBucketThreads) void bucket_small_kernel(const void* __restrict__ X, int id_dtype, int64_t x_stride,
                                                                      int col, int B, int S, int32_t* __restrict__ sid,
                                                                      int32_t* __restrict__ order, int32_t* __restrict__ seg,
                                                                      int32_t* __restrict__ status) {
    extern __shared__ int32_t s_pos[];                 // [SP][R][kBucketWaves] counts, then exclusive prefixes
    __shared__ int32_t s_wave[kBucketWaves];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int R = (B + kBucketThreads - 1) / kBucketThreads;
    const int n = SP * R * kBucketWaves;
    bool bad = false;
    // (eight rounds' ids requested together: one workgroup walks the whole batch, and a round that waits for its own strided 4-byte
    //  loads before the next round asks for its ones took 54 us at the prediction batch of 32,768 - round 6)
    constexpr int kAhead = 8;
    for (int r0 = 0; r0 < R; r0 += kAhead) {
        int64_t v8[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int b = (r0 + u) * kBucketThreads + t;
            v8[u] = (r0 + u < R && b < B) ? load_id(X, id_dtype, x_stride, b, col) : -1;
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int r = r0 + u;
            if (r >= R) break;                         // (uniform)
            const int b = r * kBucketThreads + t;
            int id = -1;                               // (beyond B: no scenario)
            if (b < B) {
                int64_t v = v8[u];
                if (v < 0 || v >= S) { bad = true; v = 0; }
                id = (int)v;
                sid[b] = id;
            }
#pragma unroll
            for (int sc = 0; sc < SP; ++sc) {
                const uint64_t mask = __ballot(id == sc);
                if (lane == 0) s_pos[(sc * R + r) * kBucketWaves + w] = __popcll(mask);
            }
        }
    }
    if (bad) atomicOr(status, 1);
    __syncthreads();
    // exclusive prefix over the n counts: thread t scans the consecutive entries [t per, (t + 1) per)
    const int per = (n + kBucketThreads - 1) / kBucketThreads;
    int32_t mine = 0;
    for (int k = t * per; k < min(n, (t + 1) * per); ++k) mine += s_pos[k];
    int32_t incl = mine;                               // inclusive scan of `mine` over the workgroup: lanes, then waves
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    int32_t run = incl - mine;
    for (int ww = 0; ww < w; ++ww) run += s_wave[ww];
    for (int k = t * per; k < min(n, (t + 1) * per); ++k) {
        const int32_t c = s_pos[k];
        s_pos[k] = run;
        run += c;
    }
    __syncthreads();
    if (t <= S) seg[t] = t == S ? B : s_pos[t * R * kBucketWaves];
    for (int r0 = 0; r0 < R; r0 += kAhead) {
        int id8[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int b = (r0 + u) * kBucketThreads + t;
            id8[u] = (r0 + u < R && b < B) ? sid[b] : -1;      // (written by this very thread above)
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int r = r0 + u;
            if (r >= R) break;
            const int b = r * kBucketThreads + t;
            const int id = id8[u];
            uint64_t my_mask = 0;
#pragma unroll
            for (int sc = 0; sc < SP; ++sc) {
                const uint64_t mask = __ballot(id == sc);
                if (id == sc) my_mask = mask;
            }
            if (b < B) order[s_pos[(id * R + r) * kBucketWaves + w] + lanes_below(my_mask)] = b;
        }
    }
}

// The same stable counting sort for a LARGE batch (the prediction batch of 32,768: one workgroup fetching 32,768 strided ids - a
// cache line each - is bound by one CU's line rate: 52 us) in two launches of one workgroup per round of 1,024 samples:
//   bucket_count_kernel    round r: ids -> sid, ballots -> counts[s][r][w]                        (global, SP x R x 16 ints)
//   bucket_place_kernel    round r: every workgroup scans the (small) count array itself - exclusive prefix in (scenario, round,
//                          wave) order, exactly bucket_small_kernel's - and places its own samples; workgroup 0 writes seg
// Same `order`, `seg`, `sid` and status as the one-workgroup kernel, bit for bit.
template <int SP>
__global__ __launch_bounds__(kBucketThreads) void bucket_count_kernel(const void* __restrict__ X, int id_dtype, int64_t x_stride,
                                                                      int col, int B, int S, int32_t* __restrict__ sid,
                                                                      int32_t* __restrict__ counts, int32_t* __restrict__ status) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, r = blockIdx.x, R = gridDim.x;
    const int b = r * kBucketThreads + t;
    int id = -1;
    if (b < B) {
        int64_t v = load_id(X, id_dtype, x_stride, b, col);
        if (v < 0 || v >= S) { atomicOr(status, 1); v = 0; }
        id = (int)v;
        sid[b] = id;
    }
#pragma unroll
    for (int sc = 0; sc < SP; ++sc) {
        const uint64_t mask = __ballot(id == sc);
        if (lane == 0) counts[(sc * R + r) * kBucketWaves + w] = __popcll(mask);
    }
}
template <int SP>
__global__ __launch_bounds__(kBucketThreads) void bucket_place_kernel(int B, int S, const int32_t* __restrict__ sid,
                                                                      const int32_t* __restrict__ counts,
                                                                      int32_t* __restrict__ order, int32_t* __restrict__ seg) {
    extern __shared__ int32_t s_pos[];                 // [SP][R][kBucketWaves] counts, then exclusive prefixes
    __shared__ int32_t s_wave[kBucketWaves];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, r = blockIdx.x, R = gridDim.x;
    const int n = SP * R * kBucketWaves;
    for (int k = t; k < n; k += kBucketThreads) s_pos[k] = counts[k];
    __syncthreads();
    const int per = (n + kBucketThreads - 1) / kBucketThreads;
    int32_t mine = 0;
    for (int k = t * per; k < min(n, (t + 1) * per); ++k) mine += s_pos[k];
    int32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    int32_t run = incl - mine;
    for (int ww = 0; ww < w; ++ww) run += s_wave[ww];
    for (int k = t * per; k < min(n, (t + 1) * per); ++k) {
        const int32_t c = s_pos[k];
        s_pos[k] = run;
        run += c;
    }
    __syncthreads();
    if (r == 0 && t <= S) seg[t] = t == S ? B : s_pos[t * R * kBucketWaves];
    const int b = r * kBucketThreads + t;
    const int id = b < B ? sid[b] : -1;
    uint64_t my_mask = 0;
#pragma unroll
    for (int sc = 0; sc < SP; ++sc) {
        const uint64_t mask = __ballot(id == sc);
        if (id == sc) my_mask = mask;
    }
    if (b < B) order[s_pos[(id * R + r) * kBucketWaves + w] + lanes_below(my_mask)] = b;
}

static int bits_for(int64_t n) {  // number of key bits needed for values in [0, n)
    int bits = 1;
    while (((int64_t)1 << bits) < n) ++bits;
    return bits;
}

struct BucketLayout {
    size_t keys_in, keys_out, idx, temp, temp_bytes, total;
};

static BucketLayout bucket_layout(int B, int S) {
    BucketLayout L{};
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t temp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, temp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
                              (int32_t*)nullptr, (unsigned)B, 0u, (unsigned)bits_for(S), (hipStream_t)0);
    L.keys_in = 0;
    L.keys_out = align(L.keys_in + sizeof(uint32_t) * (size_t)B);
    L.idx = align(L.keys_out + sizeof(uint32_t) * (size_t)B);
    L.temp = align(L.idx + sizeof(int32_t) * (size_t)B);
    L.temp_bytes = temp;
    L.total = align(L.temp + temp);
    return L;
}

}  // namespace satrans

using namespace satrans;

extern "C" int64_t satrans_bucket_workspace_bytes(int B, int S) {
    if (B <= 0 || S <= 0) return 0;
    return (int64_t)bucket_layout(B, S).total;
}

extern "C" int satrans_bucket_scenarios(const void* X, int id_dtype, int64_t x_stride, int col, int B, int S,
                                        int32_t* sid, int32_t* order, int32_t* seg, int32_t* status,
                                        void* workspace, int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(X && sid && order && seg && status && workspace, SATRANS_E_BADARG, "bucket_scenarios: null pointer");
    SATRANS_REQUIRE(B > 0 && S > 0 && col >= 0, SATRANS_E_BADARG, "bucket_scenarios: bad sizes B=%d S=%d col=%d", B, S, col);
    SATRANS_REQUIRE(id_dtype >= 0 && id_dtype <= 2, SATRANS_E_BADARG, "bucket_scenarios: id_dtype %d", id_dtype);
    if (S <= kBucketMaxS && B <= kBucketThreads * kBucketMaxRounds) {
        const int R = (int)ceil_div(B, kBucketThreads);
        const int SP = S <= 4 ? 4 : (S <= 8 ? 8 : 16);
        const size_t lds = (size_t)SP * R * kBucketWaves * sizeof(int32_t);      // <= 64 KB
        // 16 rounds and more (B > 15,360; 8 scenario rows at most: 32 KB of counts in LDS per workgroup): two launches of R
        // workgroups (-DSATRANS_EXP_BUCKET_ONE_WG: the one-workgroup kernel at every size)
#ifndef SATRANS_EXP_BUCKET_ONE_WG
        if (R >= 16 && SP <= 8 && (int64_t)lds <= workspace_bytes) {
            int32_t* counts = (int32_t*)workspace;
#define SATRANS_BUCKET2(SPV)                                                                                                    \
            bucket_count_kernel<SPV><<<R, kBucketThreads, 0, stream>>>(X, id_dtype, x_stride, col, B, S, sid, counts, status);   \
            bucket_place_kernel<SPV><<<R, kBucketThreads, lds, stream>>>(B, S, sid, counts, order, seg)
            if (SP == 4) { SATRANS_BUCKET2(4); } else { SATRANS_BUCKET2(8); }
#undef SATRANS_BUCKET2
            SATRANS_CHECK_LAUNCH("bucket_place_kernel");
            return SATRANS_OK;
        }
#endif
#define SATRANS_BUCKET(SPV)                                                                                                     \
        bucket_small_kernel<SPV><<<1, kBucketThreads, lds, stream>>>(X, id_dtype, x_stride, col, B, S, sid, order, seg, status)
        if (SP == 4) SATRANS_BUCKET(4);
        else if (SP == 8) SATRANS_BUCKET(8);
        else {
            static bool attr_set = false;
            if (!attr_set) {      // (16 scenario rows x 64 rounds: 64 KB of dynamic LDS next to the static wave sums)
                hipError_t ea = hipFuncSetAttribute((const void*)bucket_small_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    kBucketMaxS * kBucketMaxRounds * kBucketWaves * (int)sizeof(int32_t));
                SATRANS_REQUIRE(ea == hipSuccess, SATRANS_E_LAUNCH, "bucket_scenarios: LDS attribute: %s", hipGetErrorString(ea));
                attr_set = true;
            }
            SATRANS_BUCKET(16);
        }
#undef SATRANS_BUCKET
        SATRANS_CHECK_LAUNCH("bucket_small_kernel");
        return SATRANS_OK;
    }
    const BucketLayout L = bucket_layout(B, S);
    SATRANS_REQUIRE((int64_t)L.total <= workspace_bytes, SATRANS_E_WORKSPACE,
                    "bucket_scenarios: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.total);
    char* ws = (char*)workspace;
    uint32_t* keys_in = (uint32_t*)(ws + L.keys_in);
    uint32_t* keys_out = (uint32_t*)(ws + L.keys_out);
    int32_t* idx = (int32_t*)(ws + L.idx);
    const int block = 256;
    extract_scenario_kernel<<<(unsigned)ceil_div(B, block), block, 0, stream>>>(X, id_dtype, x_stride, col, B, S, sid,
                                                                              keys_in, idx, status);
    SATRANS_CHECK_LAUNCH("extract_scenario_kernel");
    size_t temp_bytes = L.temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(ws + L.temp, temp_bytes, (const uint32_t*)keys_in, keys_out,
                                             (const int32_t*)idx, order, (unsigned)B, 0u, (unsigned)bits_for(S), stream);
    SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "bucket_scenarios: radix sort: %s", hipGetErrorString(e));
    segment_bounds_kernel<<<(unsigned)ceil_div(B + 1, block), block, 0, stream>>>(keys_out, B, S, seg);
    SATRANS_CHECK_LAUNCH("segment_bounds_kernel");
    return SATRANS_OK;
}

extern "C" int satrans_gather_fwd(const float* arena, const int64_t* row_span, const int32_t* cols, const void* X,
                                  int id_dtype, int64_t x_stride, int B, int F, int D, float* out,
                                  int32_t* rows_out, int32_t* status, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(arena && row_span && cols && X && status && (out || rows_out), SATRANS_E_BADARG, "gather_fwd: null pointer");
    SATRANS_REQUIRE(B > 0 && F > 0, SATRANS_E_BADARG, "gather_fwd: bad sizes B=%d F=%d", B, F);
    SATRANS_REQUIRE(id_dtype >= 0 && id_dtype <= 2, SATRANS_E_BADARG, "gather_fwd: id_dtype %d", id_dtype);
    SATRANS_REQUIRE(D == 16 || D == 32 || D == 64 || D == 128, SATRANS_E_UNSUPPORTED,
                    "gather_fwd: embedding_dim %d not in {16,32,64,128}", D);
    const int64_t n_rows = (int64_t)B * F;
    SATRANS_REQUIRE(n_rows < ((int64_t)1 << 31), SATRANS_E_BADARG, "gather_fwd: B * F = %lld does not fit 31 bits", (long long)n_rows);
    const int lpr = D / 4;
    // enough threads for every row once, capped at 8 blocks per CU (256 CUs) and grid-strided beyond that
    int64_t blocks = ceil_div(ceil_div(n_rows, kRowsPerThread) * lpr, kGatherBlock);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
#define LAUNCH(LPR)                                                                                            \
    gather_rows_kernel<LPR><<<(unsigned)blocks, kGatherBlock, 0, stream>>>(                                    \
        (const float4*)arena, row_span, cols, X, id_dtype, x_stride, n_rows, F, (float4*)out, rows_out, status)
    switch (lpr) {
        case 4: LAUNCH(4); break;
        case 8: LAUNCH(8); break;
        case 16: LAUNCH(16); break;
        default: LAUNCH(32); break;
    }
#undef LAUNCH
    SATRANS_CHECK_LAUNCH("gather_rows_kernel");
    return SATRANS_OK;
}

// ---- measurement aid: the gather's READ side alone -----------------------------------------------------------------
// With the gather fused into the first layer no kernel writes [B,F,D]; what the fused consumer sees is this access pattern:
// arena rows by row number, D/4 lanes x 16 bytes per row, eight independent rows in flight per thread, nothing written but
// one checksum per thread.  bench.py times it to report the achieved HBM READ rate of the gather against the chip's peak
// (the copy kernel above can never exceed half of its own traffic in reads).
namespace satrans {
template <int LPR>
__global__ __launch_bounds__(kGatherBlock) void gather_read_kernel(const float4* __restrict__ arena,
                                                                 const int32_t* __restrict__ rows, int64_t n_rows,
                                                                 float* __restrict__ sink) {
    constexpr int R = 8;
    const int64_t tid = (int64_t)blockIdx.x * kGatherBlock + threadIdx.x;
    const int64_t slot0 = tid / LPR;
    const int q = (int)(tid % LPR);
    const int64_t stride = (int64_t)gridDim.x * kGatherBlock / LPR;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t base = slot0; base < n_rows; base += stride * R) {
        int32_t row[R];
        float4 val[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t slot = base + r * stride;
            row[r] = slot < n_rows ? rows[slot] : -1;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) val[r] = row[r] >= 0 ? arena[(int64_t)row[r] * LPR + q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < R; ++r) { acc.x += val[r].x; acc.y += val[r].y; acc.z += val[r].z; acc.w += val[r].w; }
    }
    sink[tid] = (acc.x + acc.y) + (acc.z + acc.w);
}
}  // namespace satrans

extern "C" int64_t satrans_gather_read_probe_floats(void) { return (int64_t)256 * 8 * satrans::kGatherBlock; }

extern "C" int satrans_gather_read_probe(const float* arena, const int32_t* rows, int64_t n_rows, int D, float* sink,
                                         void* stream_) {
    using namespace satrans;
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(arena && rows && sink && n_rows > 0, SATRANS_E_BADARG, "gather_read_probe: bad arguments");
    SATRANS_REQUIRE(D == 16 || D == 32 || D == 64 || D == 128, SATRANS_E_UNSUPPORTED, "gather_read_probe: embedding_dim %d", D);
    const int lpr = D / 4;
    int64_t blocks = ceil_div(ceil_div(n_rows, 8) * lpr, kGatherBlock);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    switch (lpr) {
        case 4: gather_read_kernel<4><<<(unsigned)blocks, kGatherBlock, 0, stream>>>((const float4*)arena, rows, n_rows, sink); break;
        case 8: gather_read_kernel<8><<<(unsigned)blocks, kGatherBlock, 0, stream>>>((const float4*)arena, rows, n_rows, sink); break;
        case 16: gather_read_kernel<16><<<(unsigned)blocks, kGatherBlock, 0, stream>>>((const float4*)arena, rows, n_rows, sink); break;
        default: gather_read_kernel<32><<<(unsigned)blocks, kGatherBlock, 0, stream>>>((const float4*)arena, rows, n_rows, sink); break;
    }
    SATRANS_CHECK_LAUNCH("gather_read_kernel");
    return SATRANS_OK;
}
