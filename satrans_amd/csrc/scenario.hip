// Per-scenario generated weights: tab[s] = relu(emb[s]) @ W^T + bias   (reference satrans.py:213,217-218 with the
// scenario encoder DNN_v2 = one Linear, submodules.py:31-61), evaluated once per SCENARIO instead of once per sample,
// and its backward.  S is a handful of rows, so both directions are latency-bound: one launch each instead of the
// dozen tiny framework kernels an autograd graph over [S,De] x [De,P] costs.
#include "common.h"

namespace satrans {

// 32 lanes per generated parameter p (coalesced 128-byte reads of its weight row), shuffle reduction per scenario
__global__ __launch_bounds__(256) void scenario_table_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ W,
                                                               const float* __restrict__ bias, int S, int De, int P,
                                                               float* __restrict__ tab) {
    const int lane = threadIdx.x & 31;
    const int p = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (p >= P) return;           // whole 32-lane groups leave together
    const float* w = W + (size_t)p * De;
    const float b = bias[p];
    for (int s = 0; s < S; ++s) {
        float acc = 0.f;
        for (int k = lane; k < De; k += 32) acc = fmaf(fmaxf(emb[(size_t)s * De + k], 0.f), w[k], acc);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 32);
        if (lane == 0) tab[(size_t)s * P + p] = acc + b;
    }
}

// g_W[p][k] += sum_s g_tab[s][p] relu(emb[s][k]);  g_bias[p] += sum_s g_tab[s][p]
__global__ __launch_bounds__(256) void scenario_table_bwd_w_kernel(const float* __restrict__ emb,
                                                                 const float* __restrict__ g_tab, int S, int De, int P,
                                                                 float* __restrict__ g_W, float* __restrict__ g_bias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // P * De fits 31 bits (checked by the caller)
    if (i >= P * De) return;
    const int p = i / De, k = i - p * De;
    float acc = 0.f, accb = 0.f;
    for (int s = 0; s < S; ++s) {
        const float g = g_tab[(size_t)s * P + p];
        acc = fmaf(g, fmaxf(emb[(size_t)s * De + k], 0.f), acc);
        accb += g;
    }
    g_W[i] += acc;
    if (k == 0) g_bias[p] += accb;
}

// g_emb[s][k] += [emb[s][k] > 0] * sum_p g_tab[s][p] W[p][k] in two fixed-order levels: block (s, slice of p) -> partial
// sums [S][kSlices][De] (8 sub-slices per block combined in order), then one thread per (s, k) adds the kSlices partials.
constexpr int kSlices = 32, kSub = 8;
__global__ __launch_bounds__(256) void scenario_table_bwd_e1_kernel(const float* __restrict__ W, const float* __restrict__ g_tab,
                                                                  int De, int P, float* __restrict__ partial) {
    extern __shared__ float s_part[];   // [kSub][De]
    const int s = blockIdx.x, slice = blockIdx.y;
    const int per = 256 / kSub;         // columns k handled per pass
    const int sub = threadIdx.x / per, kk = threadIdx.x % per;
    const int span = (P + kSlices - 1) / kSlices, sspan = (span + kSub - 1) / kSub;
    const int p0 = min(P, slice * span + sub * sspan), p1 = min(min(P, (slice + 1) * span), p0 + sspan);
    for (int k0 = 0; k0 < De; k0 += per) {
        const int k = k0 + kk;
        if (k < De) {
            float acc = 0.f;
            for (int p = p0; p < p1; ++p) acc = fmaf(g_tab[(size_t)s * P + p], W[(size_t)p * De + k], acc);
            s_part[sub * De + k] = acc;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < De; k += blockDim.x) {
        float t = 0.f;
        for (int sh = 0; sh < kSub; ++sh) t += s_part[sh * De + k];
        partial[((size_t)s * kSlices + slice) * De + k] = t;
    }
}

// The two kernels above as ONE launch (they are independent; each is at the ~5 us launch floor): blocks [0, n_w) run the weight /
// bias part, the S * kSlices blocks behind them the first level of the embedding part.
__global__ __launch_bounds__(256) void scenario_table_bwd_we1_kernel(const float* __restrict__ emb, const float* __restrict__ W,
                                                                   const float* __restrict__ g_tab, int S, int De, int P,
                                                                   float* __restrict__ g_W, float* __restrict__ g_bias,
                                                                   float* __restrict__ partial, int n_w) {
    __builtin_amdgcn_s_setprio(3);      // (ahead of the side stream's next-batch kernels when they share a SIMD: 71 -> 62 us for the touched-row chain)
    extern __shared__ float s_part[];   // [kSub][De]
    if ((int)blockIdx.x < n_w) {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= P * De) return;
        const int p = i / De, k = i - p * De;
        float acc = 0.f, accb = 0.f;
        for (int s = 0; s < S; ++s) {
            const float g = g_tab[(size_t)s * P + p];
            acc = fmaf(g, fmaxf(emb[(size_t)s * De + k], 0.f), acc);
            accb += g;
        }
        g_W[i] += acc;
        if (k == 0) g_bias[p] += accb;
        return;
    }
    const int bx = blockIdx.x - n_w;
    const int s = bx / kSlices, slice = bx - s * kSlices;
    const int per = 256 / kSub;
    const int sub = threadIdx.x / per, kk = threadIdx.x % per;
    const int span = (P + kSlices - 1) / kSlices, sspan = (span + kSub - 1) / kSub;
    const int p0 = min(P, slice * span + sub * sspan), p1 = min(min(P, (slice + 1) * span), p0 + sspan);
    for (int k0 = 0; k0 < De; k0 += per) {
        const int k = k0 + kk;
        if (k < De) {
            float acc = 0.f;
            for (int p = p0; p < p1; ++p) acc = fmaf(g_tab[(size_t)s * P + p], W[(size_t)p * De + k], acc);
            s_part[sub * De + k] = acc;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < De; k += blockDim.x) {
        float t = 0.f;
        for (int sh = 0; sh < kSub; ++sh) t += s_part[sh * De + k];
        partial[((size_t)s * kSlices + slice) * De + k] = t;
    }
}

__global__ void scenario_table_bwd_e2_kernel(const float* __restrict__ emb, const float* __restrict__ partial, int S, int De,
                                             float* __restrict__ g_emb) {
    __builtin_amdgcn_s_setprio(3);      // (ahead of the side stream's next-batch kernels when they share a SIMD: 71 -> 62 us for the touched-row chain)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * De) return;
    const int s = i / De, k = i - s * De;
    float t = 0.f;
    for (int sl = 0; sl < kSlices; ++sl) t += partial[((size_t)s * kSlices + sl) * De + k];
    if (emb[i] > 0.f) g_emb[i] += t;
}

// ---- inputs of the encoder for the variants (reference satrans.py:203-207,225-234) ---------------------------------------------
// Row (lr * S + s) of E, lr = 2 * layer + role (role 0 = Q, 1 = K; without 'pos': one lr):
//   E[.][0:D)  = mean over the C scenario columns of their embedding rows (one column: the row itself),
//   E[.][D:2D) = layerid_emb[layer] + qkvid_emb[role]                                      ('pos' only)
// The encoder applies relu to the whole row (the reference's relu(cat[relu(dom), pos]) = relu(cat[dom, pos])), so the
// table kernels above run unchanged on LR * S rows of width De = D or 2 D.
constexpr int kMaxScenarioColumns = 8;
struct ScenarioTables {
    const float* tab[kMaxScenarioColumns];
    float* g_tab[kMaxScenarioColumns];
    int rows[kMaxScenarioColumns];
};

__global__ void scenario_inputs_fwd_kernel(ScenarioTables T, const int32_t* __restrict__ index, int C, int S, int D,
                                           const float* __restrict__ lay, const float* __restrict__ role, int LR, int De,
                                           float* __restrict__ E) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= LR * S * De) return;
    const int row = i / De, k = i - row * De;
    const int lr = row / S, s = row - lr * S;
    float v;
    if (k < D) {
        v = 0.f;
        for (int c = 0; c < C; ++c) v += T.tab[c][(size_t)(index ? index[c * S + s] : s) * D + k];    // torch: stack(...).mean(-1)
        if (C > 1) v = v / (float)C;
    } else {
        v = lay[(lr >> 1) * D + (k - D)] + role[(lr & 1) * D + (k - D)];
    }
    E[i] = v;
}

// scatter of g_E (already masked by relu') back to the scenario tables and the positional embeddings, fixed order
__global__ void scenario_inputs_bwd_kernel(ScenarioTables T, const int32_t* __restrict__ index, int C, int S, int D,
                                           int LR, int De, const float* __restrict__ g_E, float* __restrict__ g_lay,
                                           float* __restrict__ g_role, int L) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int c = 0; c < C; ++c) {
        const int cnt = T.rows[c] * D;
        if (i < cnt) {
            const int r = i / D, k = i - r * D;
            float acc = 0.f;
            if (index) {
                for (int s = 0; s < S; ++s)
                    if (index[c * S + s] == r)
                        for (int lr = 0; lr < LR; ++lr) acc += g_E[(size_t)(lr * S + s) * De + k];
            } else if (r < S) {
                for (int lr = 0; lr < LR; ++lr) acc += g_E[(size_t)(lr * S + r) * De + k];
            }
            T.g_tab[c][i] += C > 1 ? acc / (float)C : acc;
            return;
        }
        i -= cnt;
    }
    if (De == D) return;
    if (i < L * D) {                       // layerid_embeddings[l][k] += sum over roles and scenarios
        const int l = i / D, k = i - l * D;
        float acc = 0.f;
        for (int r = 0; r < 2; ++r)
            for (int s = 0; s < S; ++s) acc += g_E[(size_t)((2 * l + r) * S + s) * De + D + k];
        g_lay[i] += acc;
        return;
    }
    i -= L * D;
    if (i < 2 * D) {                       // qkvid_embeddings[role][k] += sum over layers and scenarios (the V row gets none)
        const int r = i / D, k = i - r * D;
        float acc = 0.f;
        for (int l = 0; l < L; ++l)
            for (int s = 0; s < S; ++s) acc += g_E[(size_t)((2 * l + r) * S + s) * De + D + k];
        g_role[i] += acc;
    }
}

// 'onlyemb' (satrans.py:173-176): no encoder, the generated row is relu(scenario embedding of width P)
__global__ void scenario_relu_fwd_kernel(const float* __restrict__ emb, int64_t n, float* __restrict__ tab) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tab[i] = fmaxf(emb[i], 0.f);
}
__global__ void scenario_relu_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ g_tab, int64_t n,
                                         float* __restrict__ g_emb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && emb[i] > 0.f) g_emb[i] += g_tab[i];
}

}  // namespace satrans

using namespace satrans;

static int fill_tables(ScenarioTables& T, const float* const* tables, float* const* g_tables, const int32_t* rows, int C) {
    SATRANS_REQUIRE(C >= 1 && C <= kMaxScenarioColumns, SATRANS_E_UNSUPPORTED, "scenario_inputs: %d scenario columns (max %d)", C,
                    kMaxScenarioColumns);
    for (int c = 0; c < C; ++c) {
        T.tab[c] = tables ? tables[c] : nullptr;
        T.g_tab[c] = g_tables ? g_tables[c] : nullptr;
        T.rows[c] = rows ? rows[c] : 0;
    }
    return SATRANS_OK;
}

extern "C" int satrans_scenario_inputs_fwd(const float* const* tables, const int32_t* index, int C, int S, int D,
                                           const float* lay, const float* role, int L, float* E, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(tables && E && S > 0 && D > 0, SATRANS_E_BADARG, "scenario_inputs_fwd: bad arguments");
    SATRANS_REQUIRE((lay == nullptr) == (role == nullptr) && (!lay || L > 0), SATRANS_E_BADARG, "scenario_inputs_fwd: positions");
    ScenarioTables T;
    int rc = fill_tables(T, tables, nullptr, nullptr, C);
    if (rc) return rc;
    const int LR = lay ? 2 * L : 1, De = lay ? 2 * D : D;
    scenario_inputs_fwd_kernel<<<(unsigned)ceil_div((int64_t)LR * S * De, 256), 256, 0, stream>>>(T, index, C, S, D, lay, role, LR,
                                                                                              De, E);
    SATRANS_CHECK_LAUNCH("scenario_inputs_fwd_kernel");
    return SATRANS_OK;
}

extern "C" int satrans_scenario_inputs_bwd(float* const* g_tables, const int32_t* table_rows, const int32_t* index, int C, int S,
                                           int D, const float* g_E, float* g_lay, float* g_role, int L, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(g_tables && table_rows && g_E && S > 0 && D > 0, SATRANS_E_BADARG, "scenario_inputs_bwd: bad arguments");
    SATRANS_REQUIRE((g_lay == nullptr) == (g_role == nullptr) && (!g_lay || L > 0), SATRANS_E_BADARG, "scenario_inputs_bwd: positions");
    ScenarioTables T;
    int rc = fill_tables(T, nullptr, g_tables, table_rows, C);
    if (rc) return rc;
    const int LR = g_lay ? 2 * L : 1, De = g_lay ? 2 * D : D;
    int64_t total = g_lay ? (int64_t)(L + 2) * D : 0;
    for (int c = 0; c < C; ++c) total += (int64_t)table_rows[c] * D;
    scenario_inputs_bwd_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, stream>>>(T, index, C, S, D, LR, De, g_E, g_lay, g_role, L);
    SATRANS_CHECK_LAUNCH("scenario_inputs_bwd_kernel");
    return SATRANS_OK;
}

extern "C" int satrans_scenario_relu_fwd(const float* emb, int64_t n, float* tab, void* stream_) {
    SATRANS_REQUIRE(emb && tab && n > 0, SATRANS_E_BADARG, "scenario_relu_fwd: bad arguments");
    scenario_relu_fwd_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, (hipStream_t)stream_>>>(emb, n, tab);
    SATRANS_CHECK_LAUNCH("scenario_relu_fwd_kernel");
    return SATRANS_OK;
}

extern "C" int satrans_scenario_relu_bwd(const float* emb, const float* g_tab, int64_t n, float* g_emb, void* stream_) {
    SATRANS_REQUIRE(emb && g_tab && g_emb && n > 0, SATRANS_E_BADARG, "scenario_relu_bwd: bad arguments");
    scenario_relu_bwd_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, (hipStream_t)stream_>>>(emb, g_tab, n, g_emb);
    SATRANS_CHECK_LAUNCH("scenario_relu_bwd_kernel");
    return SATRANS_OK;
}

extern "C" int satrans_scenario_table_fwd(const float* emb, const float* W, const float* bias, int S, int De, int P,
                                          float* tab, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(emb && W && bias && tab, SATRANS_E_BADARG, "scenario_table_fwd: null pointer");
    SATRANS_REQUIRE(S > 0 && De > 0 && P > 0, SATRANS_E_BADARG, "scenario_table_fwd: S=%d De=%d P=%d", S, De, P);
    scenario_table_fwd_kernel<<<(unsigned)ceil_div(P, 8), 256, 0, stream>>>(emb, W, bias, S, De, P, tab);
    SATRANS_CHECK_LAUNCH("scenario_table_fwd_kernel");
    return SATRANS_OK;
}

extern "C" int64_t satrans_scenario_table_bwd_ws_floats(int S, int De) { return (int64_t)S * kSlices * De; }

extern "C" int satrans_scenario_table_bwd(const float* emb, const float* W, const float* g_tab, int S, int De, int P,
                                          float* g_emb, float* g_W, float* g_bias, float* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SATRANS_REQUIRE(emb && W && g_tab && g_emb && g_W && g_bias && workspace, SATRANS_E_BADARG, "scenario_table_bwd: null pointer");
    SATRANS_REQUIRE(S > 0 && De > 0 && P > 0 && (int64_t)P * De < ((int64_t)1 << 31), SATRANS_E_BADARG,
                    "scenario_table_bwd: S=%d De=%d P=%d", S, De, P);
    const int n_w = (int)ceil_div((int64_t)P * De, 256);
    scenario_table_bwd_we1_kernel<<<(unsigned)(n_w + S * kSlices), 256, sizeof(float) * kSub * De, stream>>>(emb, W, g_tab, S, De, P, g_W,
                                                                                                          g_bias, workspace, n_w);
    SATRANS_CHECK_LAUNCH("scenario_table_bwd_we1_kernel");
    scenario_table_bwd_e2_kernel<<<(unsigned)ceil_div(S * De, 256), 256, 0, stream>>>(emb, workspace, S, De, g_emb);
    SATRANS_CHECK_LAUNCH("scenario_table_bwd_e2_kernel");
    return SATRANS_OK;
}
