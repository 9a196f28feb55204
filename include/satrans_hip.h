/*
 * satrans_hip.h  --  C ABI of libsatrans_hip.so, the MI355X (gfx950) implementation of the
 * SATrans scenario-adaptive attention path.
 *
 * The reference (qwerfdsaplking/SATrans) is pure Python on PyTorch: it has no FFI of its own, so
 * the boundary a maintainer binds is the set of tensor-level operations its `SATrans.forward`,
 * `BaseModel.fit` and `torch.optim.Adam` perform for this path.  Each entry point below names the
 * reference lines it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer is a DEVICE pointer unless its name starts with `h_`;
 *   - the caller owns all memory, passes workspaces in, and keeps buffers alive until the stream
 *     has executed the call; nothing here allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); every kernel is enqueued on
 *     it, so calls are re-entrant per stream and capturable into a hipGraph;
 *   - return value: 0 on success, a negative SATRANS_E_* code otherwise; nothing throws across
 *     the ABI.  `satrans_last_error()` returns a static description of the last failure of the
 *     calling thread.
 *   - tensors are dense row-major fp32 unless stated; sizes are in elements.
 */
#ifndef SATRANS_HIP_H
#define SATRANS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: satrans_layer_desc gained the trailing `attn_save` field and satrans_set_layer_bwd8 left the library (round 3; the number
 *    was bumped one round late); round 4 added satrans_layer_bwd_head (a new entry point, no struct changed).  A caller built against an
 *    older header passes a shorter struct: satrans_abi_version() must be compared with this constant before any other call. */
/* 6: satrans_adam_hparams gained the trailing `arith` field (round 6). */
#define SATRANS_ABI_VERSION 6

/* error codes */
#define SATRANS_OK 0
#define SATRANS_E_BADARG (-1)      /* null pointer, negative size, unsupported dtype code          */
#define SATRANS_E_UNSUPPORTED (-2) /* shape outside what the kernels are built for (see below)     */
#define SATRANS_E_LAUNCH (-3)      /* hipLaunch / runtime error                                    */
#define SATRANS_E_WORKSPACE (-4)   /* workspace too small                                          */

/* id dtypes accepted wherever ids are read out of the input matrix X */
#define SATRANS_ID_F32 0 /* ids carried as floats and truncated, reference meta_basemodel.py:311,533-535 */
#define SATRANS_ID_I32 1
#define SATRANS_ID_I64 2

/* layer flags (bit mask) */
#define SATRANS_META_Q 1    /* 'Q' in meta_mode: MetaNet on the query projection, satrans.py:60-66     */
#define SATRANS_META_K 2    /* 'K' in meta_mode: MetaNet on the key projection,   satrans.py:67-73     */
#define SATRANS_RELU_OUT 4  /* flag 'relu': ReLU after Out_linear,                 satrans.py:91-92     */
#define SATRANS_NO_RES 8    /* att_res=False                                                           */
#define SATRANS_TRAIN 16    /* apply the four dropouts (p = drop_p)                                    */
#define SATRANS_GATE 32     /* flag 'gate': q,k *= 2*vec instead of the MetaNet,   satrans.py:61-62,68-69 */
#define SATRANS_BILINEAR 64 /* flag 'bilinear': per-head q_h @ M[s,h],             satrans.py:79-81     */
/* General path only (satrans_layer_fwd_generic / satrans_layer_bwd_generic; every other entry point refuses them): a stack of
 * layers (satrans.py:236-239: `for layer in self.domain_int_layers`) keeps its activations in the scenario-sorted order of
 * d->order between its layers instead of changing the order at both ends of every layer.
 *   X_SORTED  x (and, in the backward, dx) are rows [position in d->order][F][D]; x is not copied: the backward reads d->x again
 *   Y_SORTED  y (and, in the backward, dy) likewise */
#define SATRANS_X_SORTED 128
#define SATRANS_Y_SORTED 256

const char* satrans_last_error(void);
int satrans_abi_version(void);

/* A non-blocking HIP stream of the LOWEST priority the current device offers, for work that must yield to the caller's launch
 * stream whenever both have a kernel ready (the engine's next-batch preparation; torch only hands out normal and higher).
 * SATRANS_E_UNSUPPORTED on a device with one priority level.  The caller owns the handle. */
int satrans_stream_create_low_priority(void** stream_out);
int satrans_stream_destroy(void* stream);

/* HOST function (no GPU involved): the sample order of a shuffled epoch, out[0..n) = the permutation torch.randperm(n, generator=g)
 * returns on the CPU for a generator seeded with `seed` - what the reference's DataLoader(shuffle=True) iterates over
 * (models/meta_basemodel.py:279-280 -> torch RandomSampler).  Same Fisher-Yates pass and Mersenne-Twister draws as torch, with the
 * swap partners drawn a block ahead and prefetched (several times faster than torch's 11-75 ns per row).  `progress` (optional):
 * receives, with release ordering, the count of leading positions that are final - a consumer on another thread may read
 * out[0..*progress) while the pass is still running.  SATRANS_E_UNSUPPORTED for n >= 2^32 / 20 (torch uses another algorithm there). */
int satrans_host_randperm(uint64_t seed, int64_t n, int64_t* out, int64_t* progress);

/* ------------------------------------------------------------------------------------------------
 * Scenario bucketing.  Reads the scenario id column of X (reference satrans.py:203:
 * `X[:, feature_index[domain_col][0]].long()`), writes
 *   sid[B]     scenario row per sample (clamped into [0,S) is NOT done: an id outside [0,S) sets
 *              status[0] = 1, the caller turns that into the IndexError the reference raises),
 *   order[B]   sample indices stably grouped by scenario,
 *   seg[S+1]   start of every scenario's run inside `order`.
 * x_stride = elements between consecutive samples of X; col = column of the scenario id.
 * ---------------------------------------------------------------------------------------------- */
int64_t satrans_bucket_workspace_bytes(int B, int S);
int satrans_bucket_scenarios(const void* X, int id_dtype, int64_t x_stride, int col, int B, int S,
                             int32_t* sid, int32_t* order, int32_t* seg, int32_t* status, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-table embedding gather: reference meta_basemodel.py:533-535 (one nn.Embedding call per
 * SparseFeat) + concat_fun(axis=1) at satrans.py:211.
 *   arena      [total_rows, D] fp32: all tables back to back (any order; fields may share a table)
 *   row_span   [F][2] int64 (device): arena rows [lo, hi) of field f's table
 *   cols       [F] int32 (device): X column of field f
 *   out        [B, F, D]
 *   out        may be NULL: rows-only mode (ids -> arena rows, nothing is moved)
 *   rows_out   optional [B, F] int32: arena row of every gathered id (kept for the backward pass)
 *   status     [1] int32: set to 1 when an id falls outside its table
 * Bit-exact: every output element is a copy.
 * ---------------------------------------------------------------------------------------------- */
int satrans_gather_fwd(const float* arena, const int64_t* row_span, const int32_t* cols, const void* X,
                       int id_dtype, int64_t x_stride, int B, int F, int D, float* out,
                       int32_t* rows_out, int32_t* status, void* stream);

/* Measurement aid (bench.py): the READ side of the gather alone - arena rows by row number (the `rows_out` of
 * satrans_gather_fwd), eight rows in flight per thread, nothing written but a checksum per thread into `sink`
 * (satrans_gather_read_probe_floats() floats).  This is the access pattern of the first layer with the gather fused in. */
int64_t satrans_gather_read_probe_floats(void);
int satrans_gather_read_probe(const float* arena, const int32_t* rows, int64_t n_rows, int D, float* sink, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-scenario generated weights (reference satrans.py:213,217-218; encoder = DNN_v2 with one Linear,
 * submodules.py:31-61): tab[s] = relu(emb[s]) @ W^T + bias for the S scenario rows (not per sample), and its backward.
 *   emb [S, De], W [P, De] (nn.Linear layout), bias [P], tab / g_tab [S, P]
 *   backward ADDS into g_emb [S, De], g_W [P, De], g_bias [P]; workspace: satrans_scenario_table_bwd_ws_floats floats
 * ---------------------------------------------------------------------------------------------- */
int satrans_scenario_table_fwd(const float* emb, const float* W, const float* bias, int S, int De, int P, float* tab,
                               void* stream);
int64_t satrans_scenario_table_bwd_ws_floats(int S, int De);
int satrans_scenario_table_bwd(const float* emb, const float* W, const float* g_tab, int S, int De, int P,
                               float* g_emb, float* g_W, float* g_bias, float* workspace, void* stream);

/* Encoder inputs of the variants (reference satrans.py:203-207 several scenario columns, :167-171,225-234 flag 'pos'):
 * row (lr * S + s) of E [LR * S, De], lr = 2 * layer + role (role 0 = Q, 1 = K; LR = 2 L with positions, else 1):
 *   E[.][0:D)  = mean over the C scenario columns of tables[c][index[c * S + s]]   (index == NULL: C = 1, row s itself)
 *   E[.][D:2D) = lay[layer] + role[role]                                           (lay / role given: De = 2 D)
 * The table kernels above then run on LR * S rows.  `tables` / `g_tables` / `table_rows` are HOST arrays of C entries
 * (device pointers / row counts), `index` is a device array [C, S].  The backward ADDS g_E (the g_emb output of
 * satrans_scenario_table_bwd) into g_tables[c] [rows_c, D], g_lay [L, D] and the first two rows of g_role, in a fixed order. */
int satrans_scenario_inputs_fwd(const float* const* tables, const int32_t* index, int C, int S, int D, const float* lay,
                                const float* role, int L, float* E, void* stream);
int satrans_scenario_inputs_bwd(float* const* g_tables, const int32_t* table_rows, const int32_t* index, int C, int S, int D,
                                const float* g_E, float* g_lay, float* g_role, int L, void* stream);
/* flag 'onlyemb' (satrans.py:173-176): tab = relu(emb) elementwise over n = S * P values; backward ADDS into g_emb */
int satrans_scenario_relu_fwd(const float* emb, int64_t n, float* tab, void* stream);
int satrans_scenario_relu_bwd(const float* emb, const float* g_tab, int64_t n, float* g_emb, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One Meta_Transformer_Layer (reference satrans.py:50-100 with MetaNet submodules.py:77-103).
 * The generated MetaNet weights are passed as per-scenario tables (one row per scenario id) instead
 * of the reference's per-sample [B,P] matrices; row s = encoder(relu(domain_embeddings[s])), see
 * satrans_amd/satrans.py.  tab_q / tab_k may alias (no 'pos' flag).
 * ---------------------------------------------------------------------------------------------- */
typedef struct satrans_layer_desc {
    int32_t B, F, D, H; /* batch, fields (tokens per sample), embedding dim, heads                   */
    int32_t U;          /* MetaNet hidden width; generated row = [D*U | U*D] (meta units [D,U,D])    */
    int32_t S;          /* scenario rows in tab_q / tab_k                                            */
    int32_t flags;      /* SATRANS_* bit mask                                                        */
    int32_t layer;      /* layer index, only used to key the dropout counters                        */
    float drop_p;       /* 0.1 in the reference (satrans.py:27-28)                                   */
    uint32_t seed, step;
    int64_t tab_stride; /* elements between scenario rows of tab_q / tab_k                           */
    const float* x;     /* [B,F,D] layer input                                                       */
    const int32_t* sid; /* [B]                                                                       */
    const int32_t* order; /* [B]  from satrans_bucket_scenarios                                      */
    const int32_t* seg;   /* [S+1]                                                                   */
    const float *w_query, *w_key, *w_value; /* [D,D], y = x @ W                                      */
    const float* w_out;                     /* [D,D] nn.Linear weight, y = x @ W^T                   */
    const float *ln_g, *ln_b;               /* layer_norm                                            */
    const float *lnq_g, *lnq_b, *lnk_g, *lnk_b; /* Q_/K_meta_mlp.ffn_layer_norm                      */
    const float *tab_q, *tab_k;             /* [S, >=P]                                              */
    const int32_t* x_rows; /* NULL, or [B,F] row numbers (the `rows` output of satrans_gather_fwd): the layer then   *
                            * reads token (b,f) from x + x_rows[b*F+f]*D, i.e. `x` is the embedding arena and the    *
                            * gather is fused into the first layer (meta_basemodel.py:533-535 + satrans.py:211 never *
                            * materialise [B,F,D]); forward and backward alike                                       */
    float* attn_save;      /* NULL, or satrans_layer_attn_save_floats(d) floats: the forward leaves what the backward  *
                            * would otherwise recompute (softmax numerators, 1 / sum, dropout keep word and the       *
                            * attention output, and - since late round 4, fp32 products - the normalised MetaNet rows  *
                            * of both roles with their 1 / std, all per SORTED sample position) and a satrans_layer_bwd *
                            * on the same batch, bucket order and dropout counters reads it instead of running its attention- *
                            * forward phase, the MetaNet's second products and its LayerNorm statistics.  The layout   *
                            * is the library's own: size it with satrans_layer_attn_save_floats, never by hand.        *
                            * Fused kernels only (others ignore it)                                                    */
} satrans_layer_desc;

/* Measurement aid (bench.py: `roofline.launch_ms`).  While armed (on != 0), every launch of the fused layer kernels is issued with a
 * pair of HIP events that the dispatch itself signals with its begin / end timestamps; satrans_kernel_timing_read waits for the
 * recorded launches, returns how many there were (<= max reported; at most 512 are kept between reads) with their kind - 0 layer
 * forward, 1 layer backward, 2 last layer + head in one launch (satrans_layer_bwd_head) - and duration in ms, and forgets them.
 * satrans_kernel_timing returns the previous state.  Process-wide, not thread-safe: for a single measuring thread. */
int satrans_kernel_timing(int on);
int satrans_kernel_timing_read(int* kinds, float* ms, int max);

/* Floats of `attn_save` for this layer (B samples), 0 when the kernels that would run it do not use one. */
int64_t satrans_layer_attn_save_floats(const satrans_layer_desc* d);

/* Which implementation evaluates satrans_layer_fwd/_bwd (process-wide; initial value from SATRANS_LAYER_IMPL):
 * 0 = automatic: register-chained f32-MFMA kernels for the shapes they are built for ((D,U,H) = (32,64,4), (16,32,2),
 *     (64,16,4)), else the LDS-resident kernels with MFMA products (D, U multiples of 16), else the same kernels
 *     with scalar FMA loops (any shape);
 * 1 = LDS-resident kernels with scalar FMA loops (the "wavefront/VALU" ablation arm);
 * 2 = LDS-resident kernels with MFMA products. */
int satrans_set_layer_impl(int impl);


/* 1 when the register-chained fused kernels (csrc/layer_fused.hip) are built for this layer: (D, H) = (32, 4) or (16, 2) with the
 * MetaNet width U = 2 D, or with SATRANS_GATE / SATRANS_BILINEAR (which replace the MetaNet: satrans.py:61-64,68-71,79-81);
 * forward only also (64, 4) with U = 16.  satrans_layer_fwd / _bwd pick them on their own; callers ask in order to choose
 * between this path and the general one (satrans_layer_generic_supported). */
int satrans_layer_fused_supported(const satrans_layer_desc* d);

/* Every product of the fused layer kernels (projections, MetaNet, Out_linear; satrans.py:55-57,60-73,91) runs on
 * v_mfma_f32_16x16x4_f32: bit for bit a chain of fmaf - what torch's CPU matmul computes up to summation order.  (ABI 4 had an
 * opt-in mode that split fp32 operands into bf16 pairs, satrans_set_product_mode; retired in ABI 5: outside the fp32 tolerance on
 * trained weights - tools/experiments/README.md.) */

/* y [B,F,D]; att optional [H,B,F,F] (`normalized_att_scores`, satrans.py:87) */
int satrans_layer_fwd(const satrans_layer_desc* d, float* y, float* att, void* stream);

/* Evaluation forward of one layer with the dense products on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, fp32
 * accumulation; LayerNorm, softmax and the attention dot products in fp32; weights rounded to bf16 once per workgroup while
 * they are staged into LDS).  BASELINE.json configs[1] "bf16 forward": predict / evaluate only - no dropout (SATRANS_TRAIN
 * must be clear), no attention capture; results differ from satrans_layer_fwd by bf16 rounding (~1e-2 on the logits).
 * Built for (D,U,H) = (32,64,4) and (64,128,4). */
int satrans_layer_fwd_bf16_supported(const satrans_layer_desc* d);
int satrans_layer_fwd_bf16(const satrans_layer_desc* d, float* y, void* stream);
/* The whole stack of n layers (satrans.py:236-239: `for layer in self.domain_int_layers`) of an EVALUATION forward as one launch: a
 * tile's rows stay in LDS between the layers - read from HBM once, written once - and every layer's weight images are staged once
 * per workgroup; per layer the code of satrans_layer_fwd_bf16, i.e. the same bits.  1 <= n <= 4 layers of D = 32 with one shape,
 * one set of flags and one scenario bucketing; layer 0 reads its rows as satrans_layer_fwd_bf16 does (fused gather included), the
 * x / x_rows of the others are ignored.  y: the last layer's output [B][F][D]. */
int satrans_stack_fwd_bf16_supported(int n, const satrans_layer_desc* const* descs);
int satrans_stack_fwd_bf16(int n, const satrans_layer_desc* const* descs, float* y, void* stream);

/* General path for shapes the fused kernels are not built for (csrc/layer_generic.hip; first of all BASELINE configs[4]:
 * 64 fields, embedding_dim 64, MetaNet hidden 128): a layer as a short sequence of grouped f32-MFMA GEMM, LayerNorm and
 * attention launches over token rows kept in HBM in scenario-sorted order.  The forward SAVES its activations in `saved`
 * (satrans_layer_generic_saved_floats floats, one buffer per layer between forward and backward); the backward reads them and
 * uses `scratch` (satrans_layer_generic_scratch_floats floats, shared by all layers).  Arguments otherwise as
 * satrans_layer_fwd / satrans_layer_bwd (gradients ACCUMULATED in a fixed order; x_rows honoured).  Supported: D in
 * {16,32,64,128}, head dimension 8 or 16, U a multiple of 16 with D*U <= 8192; SATRANS_GATE or SATRANS_BILINEAR (not both):
 * the generated row is applied to q0 / k0 by an elementwise / per-head kernel, its gradient taken from the segment's z^T g.
 * satrans_set_generic_attention: attention arm of the forward - 0 automatic, 1 one lane per query row ("wavefront"),
 * 2 MFMA (F <= 64, head dimension 16), -1 back to SATRANS_GENERIC_ATTN / automatic. */
int satrans_layer_generic_supported(const satrans_layer_desc* d);
int64_t satrans_layer_generic_saved_floats(const satrans_layer_desc* d);
int64_t satrans_layer_generic_scratch_floats(const satrans_layer_desc* d);
int satrans_layer_fwd_generic(const satrans_layer_desc* d, float* y, float* att, float* saved, void* stream);
int satrans_layer_bwd_generic(const satrans_layer_desc* d, const float* dy, float* dx, const float* saved, float* scratch,
                              float* g_wq, float* g_wk, float* g_wv, float* g_wo, float* g_ln, float* g_lnq, float* g_lnk,
                              float* g_tab_q, float* g_tab_k, void* stream);
int satrans_set_generic_attention(int mode);

/* ------------------------------------------------------------------------------------------------
 * Sibling users of the same kernels, for the reference's baselines that plug the attention stack in.
 *
 * SelfAttention_Layer (models/submodules.py:178-238; `usetrans` in star.py, mmoe.py, ple.py, sharedbottom.py, adasparse.py):
 *     q,k,v = x W ; heads ; softmax(q k^T [/ sqrt(d)]) -> dropout -> @ v ; y = LayerNorm(relu(dropout(o) + x W_Res))
 * x, y, dy, dx [B,F,D] in the caller's order; W_* [D,D] (y = x @ W); g_ln [2,D] (gamma row, beta row).  Gradients are
 * ACCUMULATED (+=) in a fixed order; dx is written.  `saved` carries the forward's activations to the backward.
 * flags: SATRANS_TRAIN (dropout on), SATRANS_NO_RES (use_res=False), SATRANS_NO_SCALING (scaling=False). */
#define SATRANS_NO_SCALING 128
#define SATRANS_NO_NORM 256
typedef struct satrans_selfatt_desc {
    int32_t B, F, D, H, flags, layer;
    float drop_p;
    uint32_t seed, step;
    const float *x, *w_query, *w_key, *w_value, *w_res, *ln_g, *ln_b;
} satrans_selfatt_desc;
int64_t satrans_selfatt_saved_floats(const satrans_selfatt_desc* d);
int64_t satrans_selfatt_scratch_floats(const satrans_selfatt_desc* d);
int satrans_selfatt_fwd(const satrans_selfatt_desc* d, float* y, float* att, float* saved, void* stream);
int satrans_selfatt_bwd(const satrans_selfatt_desc* d, const float* dy, float* dx, const float* saved, float* scratch,
                        float* g_wq, float* g_wk, float* g_wv, float* g_wres, float* g_ln, void* stream);

/* MetaNet over an embedding block = BaseModel.meta_transformation (models/basemodel.py:191-199 with MetaNet,
 * models/submodules.py:64-103; `metatrans` in deepfm.py, dcn.py, ...):
 *     y = [LayerNorm](dropout(relu(x W1[s]) W2[s]) + x)      s = scenario of the sample, generated row tab[s] = [W1 D*U | W2 U*D]
 * order / seg from satrans_bucket_scenarios; flags: SATRANS_TRAIN, SATRANS_NO_NORM (MetaNet(use_norm=False)).
 * g_tab [S, tab_stride] and g_ln [2,D] are ACCUMULATED; dx is written. */
typedef struct satrans_metanet_desc {
    int32_t B, F, D, U, S, flags, layer;
    float drop_p;
    uint32_t seed, step;
    int64_t tab_stride;
    const float* x;
    const int32_t *order, *seg;
    const float *tab, *ln_g, *ln_b;
} satrans_metanet_desc;
int64_t satrans_metanet_saved_floats(const satrans_metanet_desc* d);
int64_t satrans_metanet_scratch_floats(const satrans_metanet_desc* d);
int satrans_metanet_fwd(const satrans_metanet_desc* d, float* y, float* saved, void* stream);
int satrans_metanet_bwd(const satrans_metanet_desc* d, const float* dy, float* dx, const float* saved, float* scratch,
                        float* g_tab, float* g_ln, void* stream);

/* Backward of one layer.  Recomputes the forward from d->x (same dropout counters), so nothing but
 * the layer input is kept between the passes.
 *   dy        [B,F,D] gradient of the layer output
 *   dx        [B,F,D] gradient of the layer input (written)
 *   slabs     workspace, satrans_layer_bwd_slab_floats(d) floats; per-workgroup partial weight gradients
 *   g_*       gradients of the parameters, ACCUMULATED (+=) in a fixed order (bitwise reproducible):
 *             g_wq,g_wk,g_wv,g_wo [D,D]; g_ln [2,D]; g_lnq [2,D]; g_lnk [2,D] (gamma row then beta row);
 *             g_tab_q, g_tab_k [S, tab_stride] (may alias when tab_q == tab_k).                        */
int64_t satrans_layer_bwd_slab_floats(const satrans_layer_desc* d);
int satrans_layer_bwd(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs,
                      float* g_wq, float* g_wk, float* g_wv, float* g_wo, float* g_ln, float* g_lnq,
                      float* g_lnk, float* g_tab_q, float* g_tab_k, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Head: flatten + [dense columns] + Linear(->1) + sigmoid (reference satrans.py:244-255) and, for
 * training, BCE(reduction='sum') (meta_basemodel.py:317) with its backward in the same launch.
 *   a        [B, F*D] last layer output
 *   dense    float matrix holding the DenseFeat columns (X itself when ids travel as floats), row stride
 *            dense_stride, dense_cols [n_dense] int32 (device) = its columns in feature order; may be null
 *   w        [F*D + n_dense], bias [1]
 *   prob     [B] sigmoid(logit);  logit optional [B]
 *   y        optional labels [B] (fp32).  When given:
 *              loss_sum[0] (double) += BCE sum over the batch (per-block partials added in block order),
 *              da [B,F*D] = dlogit * w[:F*D],  g_w += sum_b dlogit_b * [a_b | dense_b],  g_b += sum dlogit,
 *              scratch: satrans_head_scratch_floats(B, FD, n_dense) floats
 * ---------------------------------------------------------------------------------------------- */
int satrans_head(const float* a, const float* dense, int64_t dense_stride, const int32_t* dense_cols, int n_dense,
                 int B, int FD, const float* w, const float* bias, float* prob, float* logit, const float* y,
                 double* loss_sum, float* da, float* g_w, float* g_b, float* scratch, void* stream);

/* The same with the loss of `compile(loss=...)` (models/meta_basemodel.py:642-653, all reduction='sum' in fit):
 * SATRANS_LOSS_BCE binary_cross_entropy, SATRANS_LOSS_MSE F.mse_loss, SATRANS_LOSS_MAE F.l1_loss on the probability. */
#define SATRANS_LOSS_BCE 0
#define SATRANS_LOSS_MSE 1
#define SATRANS_LOSS_MAE 2
int satrans_head_loss(const float* a, const float* dense, int64_t dense_stride, const int32_t* dense_cols, int n_dense, int B,
                      int FD, const float* w, const float* bias, float* prob, float* logit, const float* y, double* loss_sum,
                      float* da, float* g_w, float* g_b, float* scratch, int loss_kind, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The LAST layer of a training step with the head fused in: one launch (plus two small fixed-order reductions) that replaces
 *   satrans_layer_fwd(last layer) + satrans_head_loss + satrans_layer_bwd(last layer)
 * i.e. reference satrans.py:50-100 (recomputed from the layer input, as satrans_layer_bwd does anyway), :244-255 (flatten +
 * dense columns + Linear + sigmoid), meta_basemodel.py:317 (loss, reduction='sum') and their backward.  A workgroup tile holds
 * whole samples, so the logit of a sample is a sum over rows the tile has in registers: neither the layer output [B,F,D] nor its
 * gradient is written or read, and the step runs L - 1 forward launches instead of L.
 *   d          the last layer (d->x = its input, flags / dropout counters as for its forward)
 *   h          head operands: w [F*D + n_dense], bias [1], labels [B] (fp32); dense / dense_stride / h_dense_cols as in
 *              satrans_head, except that h_dense_cols is a HOST array (the kernel takes the columns by value; n_dense <= 2);
 *              outputs prob [B], logit [B] (optional), loss_sum[0] (double) += the batch's loss sum, g_w / g_b ACCUMULATED;
 *              scratch: satrans_layer_bwd_head_scratch_floats(d, n_dense) floats
 *   dx, slabs, g_* as satrans_layer_bwd.
 * Built for the fused MetaNet shapes (D,U,H) = (32,64,4), (16,32,2) with fp32 products; satrans_layer_bwd_head_supported says
 * whether a layer / head pair can take this path (callers fall back to the three separate calls otherwise). */
typedef struct satrans_head_desc {
    const float* w;
    const float* bias;
    const float* labels;
    const float* dense;
    int64_t dense_stride;
    const int32_t* h_dense_cols; /* HOST array [n_dense] */
    int32_t n_dense, loss_kind;  /* SATRANS_LOSS_* */
    float* prob;
    float* logit;
    double* loss_sum;
    float* g_w;
    float* g_b;
    float* scratch;
} satrans_head_desc;
int satrans_layer_bwd_head_supported(const satrans_layer_desc* d, const satrans_head_desc* h);
/* satrans_stack_fwd_bf16 with the head behind it in the same launch (evaluation: satrans.py:236-255): h->w, h->bias, the dense
 * columns as above, outputs h->prob [B] and h->logit [B] (optional); labels / loss / gradients of `h` are not used.  A tile holds
 * whole samples, so the last layer's rows never leave LDS; head_kernel's summation order, i.e. the bits of satrans_head. */
int satrans_stack_fwd_bf16_head(int n, const satrans_layer_desc* const* descs, const satrans_head_desc* h, void* stream);
int64_t satrans_layer_bwd_head_scratch_floats(const satrans_layer_desc* d, int n_dense);
int satrans_layer_bwd_head(const satrans_layer_desc* d, const satrans_head_desc* h, float* dx, float* slabs, float* g_wq,
                           float* g_wk, float* g_wv, float* g_wo, float* g_ln, float* g_lnq, float* g_lnk, float* g_tab_q,
                           float* g_tab_k, void* stream);

/* The same two calls without their reduction launches, and ONE reduction launch for all layers of a step.  satrans_layer_bwd ends
 * with a fixed-order reduction of per-workgroup slabs (~10 us of a mostly idle GPU per layer); a training step can instead launch
 * its L backward kernels back to back - a slab buffer of satrans_layer_bwd_slab_floats(d) floats PER LAYER - and reduce them all,
 * together with the fused head's partial rows, in one launch on whichever stream suits it (the engine runs it on a side stream
 * underneath the touched-row optimizer kernels).  Same arithmetic and order per layer: the same bits as the per-layer calls.
 *   h_descs / h_slabs / h_grads   HOST arrays of n entries (n <= 8); the layers must share batch, bucketing, shape and flags
 *   head                          the descriptor given to satrans_layer_bwd_head_launch, or NULL
 * Fused kernels only: satrans_layer_bwd_deferred_supported says whether a layer can take this route. */
typedef struct satrans_layer_grads {
    float *g_wq, *g_wk, *g_wv, *g_wo, *g_ln, *g_lnq, *g_lnk, *g_tab_q, *g_tab_k;
} satrans_layer_grads;
int satrans_layer_bwd_deferred_supported(const satrans_layer_desc* d);
int satrans_layer_bwd_launch(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, void* stream);
int satrans_layer_bwd_head_launch(const satrans_layer_desc* d, const satrans_head_desc* h, float* dx, float* slabs, void* stream);
int satrans_layer_bwd_reduce(int n, const satrans_layer_desc* const* h_descs, float* const* h_slabs,
                             const satrans_layer_grads* h_grads, const satrans_head_desc* head, void* stream);

/* Dense elementwise steps of the optimizers `compile` accepts besides Adam (models/meta_basemodel.py:612-640: torch.optim.SGD
 * lr 0.01, Adagrad lr 0.01 eps 1e-10, RMSprop lr 0.01 alpha 0.99 eps 1e-8), over n floats with the dense gradient g:
 *   SATRANS_OPT_SGD      p -= lr g
 *   SATRANS_OPT_ADAGRAD  s += g^2 ; p -= lr g / (sqrt(s) + eps)
 *   SATRANS_OPT_RMSPROP  s = alpha s + (1 - alpha) g^2 ; p -= lr g / (sqrt(s) + eps)
 * Used with the dense table gradient of satrans_embed_grad_dense (every row: gathered rows + 2 l2 p), i.e. the reference's
 * dense semantics as one sweep over the tables per step; Adam has the lazy-exact kernels above instead. */
#define SATRANS_OPT_SGD 1
#define SATRANS_OPT_ADAGRAD 2
#define SATRANS_OPT_RMSPROP 3
int satrans_optim_flat(int kind, float* p, const float* g, float* state, int64_t n, float lr, float alpha, float eps, void* stream);
int64_t satrans_head_scratch_floats(int B, int FD, int n_dense);

/* ------------------------------------------------------------------------------------------------
 * Optimizer: torch.optim.Adam semantics (reference main.py:343) with the L2 regulariser of
 * meta_basemodel.py:577-593 folded in as its gradient 2*l2*p.
 * ---------------------------------------------------------------------------------------------- */
#define SATRANS_ADAM_EXACT 0   /* torch.optim.Adam's fp32 operations bit for bit (correctly rounded sqrt and divisions) */
#define SATRANS_ADAM_FAST  1   /* hardware sqrt / reciprocal: each update within 3.2e-7 relative of the exact one, ~2.5x fewer
                                  instructions; every kernel form (streaming, gathered rows, lazy replay / flush) runs the
                                  SAME sequence, so the forms still agree bit for bit with each other */
typedef struct satrans_adam_hparams {
    float lr_over_bc1;  /* lr / (1 - beta1^t)                */
    float bc2_sqrt;     /* sqrt(1 - beta2^t)                 */
    float beta1, beta2, eps;
    float l2;           /* l2_reg_embedding (0 for flat dense parameters) */
    int32_t arith;      /* SATRANS_ADAM_EXACT / SATRANS_ADAM_FAST (ABI 6) */
} satrans_adam_hparams;

/* Flat parameter vector (everything that is not an embedding table): p,g,m,v [n]. */
int satrans_adam_flat(float* p, const float* g, float* m, float* v, int64_t n,
                      const satrans_adam_hparams* h, void* stream);
/* the same, and out[0] += sum(vals[0..count)) in fixed order, in one launch (the step's regulariser partial sums) */
int satrans_adam_flat_sum(float* p, const float* g, float* m, float* v, int64_t n, const satrans_adam_hparams* h,
                          const double* vals, int64_t count, double* out, void* stream);

/* Embedding-gradient pipeline over n = (ranks*)B*F gathered rows.
 *   rows [n] int32 arena rows, gemb [n, D] gradient of every gathered row.
 * step 1 (sort):   sorted_rows[n], src[n] = stable sort of rows with their positions;
 *                  touched bitmap [ceil(total_rows/32)] uint32 is cleared and rebuilt (pass NULL to skip it: only
 *                  step 3 reads it).
 * step 2 (touched rows): per distinct row r: g = (sum of its gemb rows in position order) + 2*l2*p[r];
 *                  Adam on arena/m/v row r.
 * step 3 (untouched rows): every row whose bit is clear gets g = 2*l2*p (the reference's dense Adam
 *                  moves EVERY row EVERY step because of the dense regulariser gradient).
 * reg_partials [satrans_embed_reg_partials(total_rows, n, D)] doubles: per-block sums of l2*p^2 over the
 * pre-update values (reference `reg_loss`), filled by step 2 and 3 into disjoint slots.
 */
int64_t satrans_embed_sort_workspace_bytes(int64_t n, int64_t total_rows);
/* positions: optional [n] int32 with positions[i] = i kept by the caller (saves the launch that would write it) */
int satrans_embed_sort(const int32_t* rows, int64_t n, int64_t total_rows, int32_t* sorted_rows,
                       int32_t* src, uint32_t* touched, void* workspace, int64_t workspace_bytes,
                       const int32_t* positions, void* stream);
/* The same sort for the rows of one batch, rows [B, F] (position = b*F + f), when every field has a table of its own: one launch,
 * one workgroup per field sorting in LDS.  seg_field / seg_lo / seg_rows are HOST arrays of F entries: the fields in arena order
 * with the first arena row and the row count of their tables (pairwise disjoint, ascending).  B <= 8192, F <= 64, otherwise
 * SATRANS_E_UNSUPPORTED.  Output identical to satrans_embed_sort's. */
int satrans_embed_sort_fields(const int32_t* rows, int B, int F, const int32_t* seg_field, const int32_t* seg_lo,
                              const int32_t* seg_rows, int32_t* sorted_rows, int32_t* src, void* stream);
/* ... and with the ids -> rows translation of satrans_gather_fwd(out = NULL, rows_out = rows) in the same launch (meta_basemodel.py:
 * 533-535, the lookup's index arithmetic): `rows` [B, F] is an OUTPUT; X / id_dtype / x_stride / cols / row_span / status as for
 * satrans_gather_fwd (an id outside its table sets bit 0 of *status and is recorded as the table's first row); seg_lo[k] must be
 * row_span[2 * seg_field[k]]. */
int satrans_embed_rows_sort_fields(const void* X, int id_dtype, int64_t x_stride, const int32_t* cols, const int64_t* row_span,
                                   int32_t* rows, int B, int F, const int32_t* seg_field, const int32_t* seg_lo,
                                   const int32_t* seg_rows, int32_t* sorted_rows, int32_t* src, int32_t* status, void* stream);
/* The same sort when `ids` [n] is W sorted runs back to back (owner form of the data-parallel step: what W ranks sent an owner;
 * each run ascending, equal rows in position order): ONE launch, every element ranks itself by W - 1 binary searches.  h_run_start:
 * HOST array of W + 1 boundaries (h_run_start[0] = 0, h_run_start[W] = n), W <= 64.  Output identical to
 * satrans_embed_sort(ids, positions = 0..n-1): src[j] = the index in `ids` of sorted element j. */
int satrans_embed_merge_runs(const int32_t* ids, int64_t n, const int64_t* h_run_start, int W, int32_t* sorted_rows, int32_t* src,
                             void* stream);
/* inv[src[i]] = i for i < n: the inverse of a sort's source positions */
int satrans_embed_inverse_positions(const int32_t* src, int64_t n, int32_t* inv, void* stream);
int64_t satrans_embed_reg_partials(int64_t total_rows, int64_t n, int D);
/* last / t: optional (lazy form): last[row] = t for every row stepped */
int satrans_embed_adam_touched(float* arena, float* m, float* v, int D, const int32_t* sorted_rows,
                               const int32_t* src, int64_t n, const float* gemb, float* partial_ws,
                               const satrans_adam_hparams* h, double* reg_partials, int32_t* last, int t,
                               void* stream);
int64_t satrans_embed_partial_ws_floats(int64_t n, int D);
/* rows [first_row, total_rows) that are not in the bitmap; grid_blocks: 0 = the measured optimum (512 persistent blocks
 * of 256 threads), otherwise the grid size to use */
int satrans_embed_adam_untouched(float* arena, float* m, float* v, int64_t first_row, int64_t total_rows, int D,
                                 const uint32_t* touched, const satrans_adam_hparams* h,
                                 double* reg_partials, int grid_blocks, void* stream);
/* touched-row bitmap [ceil(total_rows/32)] of an already sorted id list (cleared first; n may be 0) */
int satrans_embed_mark_touched(const int32_t* sorted_rows, int64_t n, int64_t total_rows, uint32_t* touched,
                               void* stream);

/* Small tables (the caller places them first in the arena): their gradient is a DENSE buffer g_rows [rows, D] - the
 * ordered segmented sums of satrans_embed_segment_sums over the sorted positions that fall into those tables, stored
 * instead of applied (rows nobody gathered keep the zeros the caller wrote) - which data-parallel ranks all-reduce, and
 * satrans_embed_adam_rows then steps EVERY row of [row0, row0+rows) with g = g_rows + 2*l2*p (for an ungathered row that
 * is exactly the reference's regulariser-only step) and sets last[row] = t when `last` is given.
 * Workspaces as for satrans_embed_adam_touched; reg_partials of adam_rows: satrans_embed_adam_rows_partials doubles.
 * satrans_embed_pack_rows: out[j] = gemb[src[j]], the gradient rows of a sorted id list in sorted order (what a rank
 * contributes to the all-gather of the large tables' gradient rows). */
int satrans_embed_segment_sums(const int32_t* sorted_rows, const int32_t* src, int64_t n, const float* gemb, int D,
                               float* partial_ws, double* reg_partials, float* g_rows, void* stream);
int64_t satrans_embed_adam_rows_partials(int64_t rows, int D);
int satrans_embed_adam_rows(float* arena, float* m, float* v, int32_t* last, int64_t row0, int64_t rows, int D,
                            const float* g_rows, const satrans_adam_hparams* h, int t, double* reg_partials,
                            void* stream);
int satrans_embed_pack_rows(const int32_t* src, int64_t n, const float* gemb, int D, float* out, void* stream);

/* Lazy-exact form of the dense step (same tables bit for bit, HBM traffic only for touched rows): instead of
 * satrans_embed_adam_untouched every step, the regulariser-only Adam steps of a row are postponed and replayed with
 * identical arithmetic when the row is next gathered (replay, over the distinct rows of the sorted ids, BEFORE the
 * gather of that step) or for all rows (flush: epoch end, before predict / state_dict).
 *   last   [total_rows] int32: last step applied to each row (0 initially)
 *   table  [>= target+1][2] fp64: table[s] = (fp32(lr / (1 - beta1^s)), 1 / (double)fp32(sqrt(1 - beta2^s))), filled by
 *          the host (the kernels divide by the per-step constant through its double reciprocal, exactly: embed_adam.hip)
 *   h      beta1, beta2, eps, l2 (lr_over_bc1 / bc2_sqrt are ignored)
 *   reg_partials [satrans_embed_lazy_reg_partials(n, D)] doubles (zero-initialised by the caller): per-block sums of
 *          l2*p^2 over the replayed steps; replay writes the first ceil(n*D/256) slots, flush the 4096 after them.
 * satrans_embed_lazy_mark sets last[r] = t for the distinct rows of a step after their Adam update. */
int64_t satrans_embed_lazy_reg_partials(int64_t n, int D);
int satrans_embed_lazy_replay(float* arena, float* m, float* v, int32_t* last, int D, const int32_t* sorted_rows,
                              int64_t n, int target, const double* table, const satrans_adam_hparams* h,
                              double* reg_partials, void* stream);
int satrans_embed_lazy_flush(float* arena, float* m, float* v, int32_t* last, int64_t total_rows, int D, int target,
                             const double* table, const satrans_adam_hparams* h, int64_t n, double* reg_partials,
                             void* stream);
int satrans_embed_lazy_mark(const int32_t* sorted_rows, int64_t n, int32_t* last, int t, void* stream);

/* Diagnostic behind the replay's arithmetic.  Replay and flush run four elements per lane on the packed fp32 pipe with the
 * square root and the division written out as correctly rounded fma sequences (embed_adam.hip); this call counts the results
 * that differ from the IEEE operations: mode 0 = square root over the floats with bit patterns [first, first + count),
 * restricted to the operand range of the packed path; mode 1 = division over `count` pseudo-random pairs (seed `first`) of that
 * range; mode 2 / 3 = the same two operations BELOW that range through their exact power-of-two scaling (every positive float
 * under 2^-100, subnormals included; numerators from the smallest subnormal to 2^-80 and signed zeros).  *mismatches is HOST
 * memory.  Used by tests/test_gpu_parity.py (every float of the range for modes 0 and 2). */
int satrans_debug_check_packed_math(int mode, uint64_t first, uint64_t count, uint64_t* mismatches, void* stream);

/* Dense materialisation of the embedding gradient (debug / parity tests only):
 * g_arena [total_rows, D] += scatter of gemb by rows (position order), + 2*l2*p when l2 != 0. */
int satrans_embed_grad_dense(const float* arena, const int32_t* sorted_rows, const int32_t* src, int64_t n,
                             const float* gemb, int64_t total_rows, int D, float l2, float* g_arena,
                             void* stream);

/* Sum of `count` doubles in index order -> out[0] (+= when accumulate != 0). */
int satrans_sum_f64(const double* v, int64_t count, double* out, int accumulate, void* stream);

/* The per-step training metrics of fit(verbose > 0) (meta_basemodel.py:330-337: sklearn log_loss and roc_auc_score on host
 * copies of every batch) for one batch in one launch: out[0] = log_loss(y, p.astype(float64)) (probabilities clipped to
 * [eps, 1 - eps] with the fp64 eps, mean), out[1] = roc_auc_score(y, p) (tied scores count one half; NaN when only one class
 * is present).  y, p: [n] fp32, n <= 8192; out: two doubles in device memory.  Fixed summation order. */
int satrans_batch_metrics(const float* y, const float* p, int n, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SATRANS_HIP_H */
