// K6 head: flatten + dense columns + Linear(F*D + n_dense -> 1) + sigmoid (reference models/satrans.py:244-255),
// fused for training with BCE(reduction='sum') (models/meta_basemodel.py:317) and its backward.
//
// HBM-bound on the [B, F*D] activation (read once; written once more as its gradient in training).  One wave
// owns one sample row at a time: 16-byte loads, wave-shuffle reduction, no LDS in the forward part.  The weight
// gradient is a reduction over the batch: every block reduces its kSamplesPerBlock samples in index order into a
// partial row, and a second kernel adds the partial rows in block order (bitwise reproducible).
#include "common.h"

namespace satrans {

constexpr int kHeadBlock = 256;
constexpr int kSamplesPerBlock = 8;   // 1024 blocks at B = 8192: the kernel is latency-bound, not bandwidth-bound

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(kHeadBlock) void head_kernel(const float* __restrict__ a, const float* __restrict__ dense,
                                                        int64_t dense_stride, const int32_t* __restrict__ dense_cols,
                                                        int n_dense, int B, int FD, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ prob,
                                                        float* __restrict__ logit, const float* __restrict__ y,
                                                        float* __restrict__ da, float* __restrict__ partial, int loss_kind) {
    __shared__ float s_dlogit[kSamplesPerBlock];
    __shared__ float s_loss[kSamplesPerBlock];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = blockIdx.x * kSamplesPerBlock;
    const int nb = min(kSamplesPerBlock, B - b0);
    const int FD4 = FD >> 2;  // FD is a multiple of 4 (D is)
    const float4* w4 = (const float4*)w;
    for (int ls = wave; ls < nb; ls += kHeadBlock / 64) {
        const int b = b0 + ls;
        const float4* row = (const float4*)(a + (size_t)b * FD);
        float acc = 0.f;
        for (int i = lane; i < FD4; i += 64) {
            const float4 xv = row[i], wv = w4[i];
            acc = fmaf(xv.x, wv.x, acc);
            acc = fmaf(xv.y, wv.y, acc);
            acc = fmaf(xv.z, wv.z, acc);
            acc = fmaf(xv.w, wv.w, acc);
        }
        for (int j = lane; j < n_dense; j += 64)
            acc = fmaf(dense[(size_t)b * dense_stride + dense_cols[j]], w[FD + j], acc);
        const float z = wave_sum(acc) + bias[0];
        const float p = 1.0f / (1.0f + expf(-z));
        if (lane == 0) {
            prob[b] = p;
            if (logit) logit[b] = z;
            if (y) {
                const float t = y[b];
                const float pq = (1.0f - p) * p;
                if (loss_kind == SATRANS_LOSS_MSE) {            // F.mse_loss(reduction='sum'): (p - t)^2, d/dp = 2 (p - t)
                    s_loss[ls] = (p - t) * (p - t);
                    s_dlogit[ls] = 2.0f * (p - t) * pq;
                } else if (loss_kind == SATRANS_LOSS_MAE) {     // F.l1_loss(reduction='sum'): |p - t|, d/dp = sign(p - t)
                    s_loss[ls] = fabsf(p - t);
                    s_dlogit[ls] = (p > t ? 1.0f : (p < t ? -1.0f : 0.0f)) * pq;
                } else {
                    // torch.nn.functional.binary_cross_entropy clamps both logs at -100
                    const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.0f - p), -100.f);
                    s_loss[ls] = -(t * lp + (1.0f - t) * lq);
                    // BCE backward (grad * (p - t) / max(p(1-p), 1e-12)) chained with sigmoid backward (* p(1-p))
                    s_dlogit[ls] = (p - t) / fmaxf(pq, 1e-12f) * pq;
                }
            }
        }
    }
    if (!y) return;
    __syncthreads();
    // da = dlogit * w (coalesced), and this block's partial of g_w / g_b / loss
    const int ncol = FD + n_dense;
    float* prow = partial + (size_t)blockIdx.x * (ncol + 2);
    for (int c = threadIdx.x; c < ncol; c += kHeadBlock) {
        float acc = 0.f;
        const float wc = w[c];
        for (int ls = 0; ls < nb; ++ls) {
            const int b = b0 + ls;
            const float g = s_dlogit[ls];
            if (c < FD) {
                acc = fmaf(g, a[(size_t)b * FD + c], acc);
                da[(size_t)b * FD + c] = g * wc;
            } else {
                acc = fmaf(g, dense[(size_t)b * dense_stride + dense_cols[c - FD]], acc);
            }
        }
        prow[c] = acc;
    }
    if (threadIdx.x == 0) {
        float gb = 0.f, ls_sum = 0.f;
        for (int ls = 0; ls < nb; ++ls) {
            gb += s_dlogit[ls];
            ls_sum += s_loss[ls];
        }
        prow[ncol] = gb;
        prow[ncol + 1] = ls_sum;
    }
}

// block = 32 columns x kReduceGroups (common.h) groups of partial rows (contiguous shares, index order), group sums combined in group order
__global__ __launch_bounds__(32 * kReduceGroups) void head_reduce_kernel(const float* __restrict__ partial, int nblk, int ncol,
                                                                       float* __restrict__ g_w, float* __restrict__ g_b,
                                                                       double* __restrict__ loss_sum) {
    __shared__ double s_acc[kReduceGroups][32];
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + lane;
    const int share = (nblk + kReduceGroups - 1) / kReduceGroups;
    const int lo = grp * share, hi = min(nblk, lo + share);
    double acc = 0.0;
    if (c < ncol + 2) {
        if (c == ncol + 1) {
            for (int k = lo; k < hi; ++k) acc += (double)partial[(size_t)k * (ncol + 2) + c];
        } else {
            float a32 = 0.f;
#pragma unroll 8                   // (the loads of eight rows in flight; the adds stay in index order)
            for (int k = lo; k < hi; ++k) a32 += partial[(size_t)k * (ncol + 2) + c];
            acc = (double)a32;
        }
    }
    s_acc[grp][lane] = acc;
    __syncthreads();
    if (grp != 0 || c >= ncol + 2) return;
    if (c == ncol + 1) {
        double t = 0.0;
        for (int k = 0; k < kReduceGroups; ++k) t += s_acc[k][lane];
        loss_sum[0] += t;
        return;
    }
    float t = 0.f;
    for (int k = 0; k < kReduceGroups; ++k) t += (float)s_acc[k][lane];
    if (c < ncol) g_w[c] += t;
    else g_b[0] += t;
}

}  // namespace satrans

using namespace satrans;

// internal: add `nblk` partial rows [g_w (ncol) | g_b | loss] in row order (the fused last-layer step writes one per workgroup)
extern "C" int satrans_head_reduce_partials(const float* partial, int nblk, int ncol, float* g_w, float* g_b, double* loss_sum,
                                            void* stream_) {
    head_reduce_kernel<<<(unsigned)ceil_div(ncol + 2, 32), 32 * kReduceGroups, 0, (hipStream_t)stream_>>>(partial, nblk, ncol, g_w, g_b,
                                                                                                    loss_sum);
    SATRANS_CHECK_LAUNCH("head_reduce_kernel");
    return SATRANS_OK;
}

extern "C" int64_t satrans_head_scratch_floats(int B, int FD, int n_dense) {
    if (B <= 0) return 0;
    return ceil_div(B, kSamplesPerBlock) * (int64_t)(FD + n_dense + 2);
}

extern "C" int satrans_head_loss(const float* a, const float* dense, int64_t dense_stride, const int32_t* dense_cols,
                                 int n_dense, int B, int FD, const float* w, const float* bias, float* prob, float* logit,
                                 const float* y, double* loss_sum, float* da, float* g_w, float* g_b, float* scratch,
                                 int loss_kind, void* stream_);

extern "C" int satrans_head(const float* a, const float* dense, int64_t dense_stride, const int32_t* dense_cols,
                            int n_dense, int B, int FD, const float* w, const float* bias, float* prob, float* logit,
                            const float* y, double* loss_sum, float* da, float* g_w, float* g_b, float* scratch,
                            void* stream_) {
    return satrans_head_loss(a, dense, dense_stride, dense_cols, n_dense, B, FD, w, bias, prob, logit, y, loss_sum, da, g_w, g_b,
                             scratch, SATRANS_LOSS_BCE, stream_);
}

extern "C" int satrans_head_loss(const float* a, const float* dense, int64_t dense_stride, const int32_t* dense_cols,
                                 int n_dense, int B, int FD, const float* w, const float* bias, float* prob, float* logit,
                                 const float* y, double* loss_sum, float* da, float* g_w, float* g_b, float* scratch,
                                 int loss_kind, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(a && w && bias && prob, SATRANS_E_BADARG, "head: null pointer");
    SATRANS_REQUIRE(B > 0 && FD > 0 && (FD % 4) == 0 && n_dense >= 0, SATRANS_E_BADARG, "head: bad sizes B=%d FD=%d", B, FD);
    SATRANS_REQUIRE(n_dense == 0 || (dense && dense_cols), SATRANS_E_BADARG, "head: dense columns without a float matrix");
    SATRANS_REQUIRE(loss_kind >= SATRANS_LOSS_BCE && loss_kind <= SATRANS_LOSS_MAE, SATRANS_E_BADARG, "head: loss kind %d", loss_kind);
    if (y) SATRANS_REQUIRE(loss_sum && da && g_w && g_b && scratch, SATRANS_E_BADARG, "head: training outputs missing");
    const int nblk = (int)ceil_div(B, kSamplesPerBlock);
    head_kernel<<<nblk, kHeadBlock, 0, stream>>>(a, dense, dense_stride, dense_cols, n_dense, B, FD, w, bias, prob, logit,
                                                  y, da, scratch, loss_kind);
    SATRANS_CHECK_LAUNCH("head_kernel");
    if (y) {
        const int ncol = FD + n_dense;
        head_reduce_kernel<<<(unsigned)ceil_div(ncol + 2, 32), 32 * kReduceGroups, 0, stream>>>(scratch, nblk, ncol, g_w, g_b, loss_sum);
        SATRANS_CHECK_LAUNCH("head_reduce_kernel");
    }
    return SATRANS_OK;
}
