// Per-step training metrics of fit(verbose > 0): sklearn.metrics.log_loss and roc_auc_score of ONE batch in ONE launch.
//
// Reference: models/meta_basemodel.py:330-337 computes them with sklearn on host copies of every batch
// (`metric_fun(y.cpu().numpy(), y_pred.cpu().numpy().astype("float64"))`).  satrans_amd/device_metrics.py restates them with
// torch ops on the device (no host sync) - a dozen small launches per step whose HOST cost (~1.3 ms) is more than the whole
// training step: fit() ran host-bound at 5.1 M samples/s beside a 6.6 M samples/s step.  Here one workgroup does a batch:
//   log_loss  -mean(y log p + (1 - y) log(1 - p)) in fp64, p clipped to [eps, 1 - eps] with the fp64 eps (sklearn);
//   roc_auc   the Mann-Whitney statistic with tied scores counted one half: scores sorted in LDS (rocPRIM block radix sort),
//             a score's tie group located by binary search in the sorted row (as device_metrics.roc_auc does with searchsorted),
//             average ranks summed over the positives in fp64 (half-integers below 2^53: exact in any order);
//             0 / 0 = NaN when only one class is present (sklearn raises there; fit() raises at the epoch's end).
// Sums are combined in a fixed order: bitwise reproducible.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace satrans {

constexpr int kMetricBlock = 1024;

template <int ITEMS>
__global__ __launch_bounds__(kMetricBlock) void batch_metrics_kernel(const float* __restrict__ y, const float* __restrict__ p, int n,
                                                                      double* __restrict__ out) {
    using Sort = rocprim::block_radix_sort<uint32_t, kMetricBlock, ITEMS, uint32_t>;
    __shared__ typename Sort::storage_type sort_storage;
    __shared__ uint32_t s_key[kMetricBlock * ITEMS];
    __shared__ double s_red[3][kMetricBlock / 64];
    uint32_t keys[ITEMS], vals[ITEMS];
    double ll = 0.0;
    constexpr double eps = 2.220446049250313e-16;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int j = (int)threadIdx.x * ITEMS + i;
        const bool in = j < n;
        const float pj = in ? p[j] : 0.f, yj = in ? y[j] : 0.f;
        const uint32_t bits = __float_as_uint(pj);
        keys[i] = in ? (bits ^ ((bits >> 31) ? 0xFFFFFFFFu : 0x80000000u)) : 0xFFFFFFFFu;      // ascending float order; padding last
        vals[i] = yj != 0.f ? 1u : 0u;
        if (in) {
            const double pc = fmin(fmax((double)pj, eps), 1.0 - eps);
            ll += (double)yj * log(pc) + (1.0 - (double)yj) * log(1.0 - pc);
        }
    }
    Sort().sort(keys, vals, sort_storage);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) s_key[(int)threadIdx.x * ITEMS + i] = keys[i];
    __syncthreads();
    double rank_pos = 0.0, n_pos = 0.0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int j = (int)threadIdx.x * ITEMS + i;
        if (j < n && vals[i]) {
            const uint32_t k = keys[i];
            int lo = 0, hi = j;                       // first position holding k: in [0, j]
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_key[mid] < k) lo = mid + 1; else hi = mid; }
            const int left = lo;
            lo = j + 1; hi = n;                       // first position beyond the tie group: in [j + 1, n]
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_key[mid] <= k) lo = mid + 1; else hi = mid; }
            rank_pos += (double)(left + lo + 1) * 0.5;   // average 1-based rank of the group [left, lo)
            n_pos += 1.0;
        }
    }
    // wave sums, then the waves in order
    double v[3] = {ll, rank_pos, n_pos};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off, 64);
        if ((threadIdx.x & 63) == 0) s_red[q][threadIdx.x >> 6] = v[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[3] = {0.0, 0.0, 0.0};
        for (int w = 0; w < kMetricBlock / 64; ++w)
            for (int q = 0; q < 3; ++q) t[q] += s_red[q][w];
        const double np_ = t[2], nn_ = (double)n - t[2];
        out[0] = -t[0] / (double)n;
        out[1] = (t[1] - np_ * (np_ + 1.0) * 0.5) / (np_ * nn_);
    }
}

}  // namespace satrans

using namespace satrans;

// out[0] = log_loss(y, p), out[1] = roc_auc_score(y, p) of one batch, n <= 8192 (fp64, device memory).  y, p: [n] fp32.
extern "C" int satrans_batch_metrics(const float* y, const float* p, int n, double* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(y && p && out, SATRANS_E_BADARG, "batch_metrics: null pointer");
    SATRANS_REQUIRE(n > 0 && n <= 8192, SATRANS_E_UNSUPPORTED, "batch_metrics: n=%d (1 .. 8192 per launch)", n);
    if (n <= 1024) batch_metrics_kernel<1><<<1, kMetricBlock, 0, stream>>>(y, p, n, out);
    else if (n <= 2048) batch_metrics_kernel<2><<<1, kMetricBlock, 0, stream>>>(y, p, n, out);
    else if (n <= 4096) batch_metrics_kernel<4><<<1, kMetricBlock, 0, stream>>>(y, p, n, out);
    else batch_metrics_kernel<8><<<1, kMetricBlock, 0, stream>>>(y, p, n, out);
    SATRANS_CHECK_LAUNCH("batch_metrics_kernel");
    return SATRANS_OK;
}
