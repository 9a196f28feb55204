// Error reporting and the dispatch of the public layer entry points over the three implementations.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace satrans {
static thread_local char g_error[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
static int g_layer_impl = -1;
}  // namespace satrans

extern "C" {

int satrans_layer_validate(const satrans_layer_desc* d, const char* who);
int satrans_layer_fused_supported(const satrans_layer_desc* d);
int satrans_layer_fwd_fused(const satrans_layer_desc* d, float* y, float* att, void* stream);
int satrans_layer_fwd_lds(const satrans_layer_desc* d, float* y, float* att, void* stream);
int satrans_layer_bwd_fused_supported(const satrans_layer_desc* d);
int64_t satrans_layer_bwd_slab_floats_fused(const satrans_layer_desc* d);
int64_t satrans_layer_attn_save_floats_fused(const satrans_layer_desc* d);
int satrans_layer_bwd_fused(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, float* g_wq, float* g_wk,
                            float* g_wv, float* g_wo, float* g_ln, float* g_lnq, float* g_lnk, float* g_tab_q,
                            float* g_tab_k, void* stream);
int64_t satrans_layer_bwd_slab_floats_lds(const satrans_layer_desc* d);
int satrans_layer_bwd_lds(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, float* g_wq, float* g_wk,
                          float* g_wv, float* g_wo, float* g_ln, float* g_lnq, float* g_lnk, float* g_tab_q,
                          float* g_tab_k, void* stream);

int satrans_layer_bwd_head_fused_supported(const satrans_layer_desc* d, const satrans_head_desc* h);
int64_t satrans_layer_bwd_head_scratch_floats_fused(const satrans_layer_desc* d, int n_dense);
int satrans_layer_bwd_head_fused(const satrans_layer_desc* d, const satrans_head_desc* h, float* dx, float* slabs, float* g_wq,
                                 float* g_wk, float* g_wv, float* g_wo, float* g_ln, float* g_lnq, float* g_lnk, float* g_tab_q,
                                 float* g_tab_k, void* stream);

int satrans_layer_bwd_launch_fused(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, void* stream);
int satrans_layer_bwd_head_launch_fused(const satrans_layer_desc* d, const satrans_head_desc* h, float* dx, float* slabs, void* stream);
int satrans_layer_bwd_reduce_fused(int n, const satrans_layer_desc* const* descs, float* const* slabs,
                                   const satrans_layer_grads* grads, const satrans_head_desc* head, void* stream);

const char* satrans_last_error(void) { return satrans::g_error; }
int satrans_abi_version(void) { return SATRANS_ABI_VERSION; }

// A non-blocking HIP stream of the LOWEST priority the current device offers (host frameworks only hand out normal and higher).
// SATRANS_E_UNSUPPORTED when the device has a single priority level.  The caller owns the handle (satrans_stream_destroy).
int satrans_stream_create_low_priority(void** out) {
    SATRANS_REQUIRE(out, SATRANS_E_BADARG, "stream_create_low_priority: null pointer");
    *out = nullptr;
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "stream_create_low_priority: %s", hipGetErrorString(e));
    SATRANS_REQUIRE(least > greatest, SATRANS_E_UNSUPPORTED, "stream_create_low_priority: one priority level only");   // (numerically larger = lower)
    hipStream_t st = nullptr;
    e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least);
    SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "stream_create_low_priority: %s", hipGetErrorString(e));
    *out = (void*)st;
    return SATRANS_OK;
}

int satrans_stream_destroy(void* stream) {
    if (!stream) return SATRANS_OK;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    SATRANS_REQUIRE(e == hipSuccess, SATRANS_E_LAUNCH, "stream_destroy: %s", hipGetErrorString(e));
    return SATRANS_OK;
}

// 0 = automatic: register-chained MFMA kernels (layer_fused.hip) for the shapes they are built for, else the LDS
//     kernels with MFMA products, else the LDS kernels with scalar FMA loops;
// 1 = LDS kernels, scalar FMA loops;  2 = LDS kernels, MFMA products.     Initial value from SATRANS_LAYER_IMPL.
int satrans_layer_impl(void) {
    if (satrans::g_layer_impl < 0) {
        const char* env = getenv("SATRANS_LAYER_IMPL");
        satrans::g_layer_impl = env ? atoi(env) : 0;
        if (satrans::g_layer_impl < 0 || satrans::g_layer_impl > 2) satrans::g_layer_impl = 0;
    }
    return satrans::g_layer_impl;
}

int satrans_set_layer_impl(int impl) {
    SATRANS_REQUIRE(impl >= 0 && impl <= 2, SATRANS_E_BADARG, "set_layer_impl: %d is not 0, 1 or 2", impl);
    satrans::g_layer_impl = impl;
    return SATRANS_OK;
}

int satrans_layer_fwd(const satrans_layer_desc* d, float* y, float* att, void* stream) {
    int rc = satrans_layer_validate(d, "layer_fwd");
    if (rc) return rc;
    SATRANS_REQUIRE(!(d->flags & (SATRANS_X_SORTED | SATRANS_Y_SORTED)), SATRANS_E_UNSUPPORTED,
                    "layer_fwd: SATRANS_X_SORTED / SATRANS_Y_SORTED are flags of the general path (satrans_layer_fwd_generic)");
    if (satrans_layer_impl() == 0 && satrans_layer_fused_supported(d)) return satrans_layer_fwd_fused(d, y, att, stream);
    return satrans_layer_fwd_lds(d, y, att, stream);
}

// large enough for whichever implementation satrans_layer_bwd may pick (the choice can change between calls through
// satrans_set_layer_impl)
int64_t satrans_layer_bwd_slab_floats(const satrans_layer_desc* d) {
    if (satrans_layer_validate(d, "layer_bwd")) return -1;
    int64_t n = satrans_layer_bwd_slab_floats_lds(d);
    if (satrans_layer_bwd_fused_supported(d)) {
        const int64_t f = satrans_layer_bwd_slab_floats_fused(d);
        if (f > n) n = f;
    }
    return n;
}

// what satrans_layer_fwd leaves for satrans_layer_bwd in d->attn_save: only the fused kernels use it
int64_t satrans_layer_attn_save_floats(const satrans_layer_desc* d) {
    if (satrans_layer_validate(d, "layer_attn_save")) return -1;
    if (satrans_layer_impl() != 0 || !satrans_layer_fused_supported(d) || !satrans_layer_bwd_fused_supported(d)) return 0;
    return satrans_layer_attn_save_floats_fused(d);
}

int satrans_layer_bwd(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, float* g_wq, float* g_wk,
                      float* g_wv, float* g_wo, float* g_ln, float* g_lnq, float* g_lnk, float* g_tab_q, float* g_tab_k,
                      void* stream) {
    int rc = satrans_layer_validate(d, "layer_bwd");
    if (rc) return rc;
    SATRANS_REQUIRE(!(d->flags & (SATRANS_X_SORTED | SATRANS_Y_SORTED)), SATRANS_E_UNSUPPORTED,
                    "layer_bwd: SATRANS_X_SORTED / SATRANS_Y_SORTED are flags of the general path (satrans_layer_bwd_generic)");
    if (satrans_layer_impl() == 0 && satrans_layer_bwd_fused_supported(d))
        return satrans_layer_bwd_fused(d, dy, dx, slabs, g_wq, g_wk, g_wv, g_wo, g_ln, g_lnq, g_lnk, g_tab_q, g_tab_k, stream);
    return satrans_layer_bwd_lds(d, dy, dx, slabs, g_wq, g_wk, g_wv, g_wo, g_ln, g_lnq, g_lnk, g_tab_q, g_tab_k, stream);
}

// the last layer of a training step with the head fused in: fused kernels only (layer_fused.hip)
int satrans_layer_bwd_head_supported(const satrans_layer_desc* d, const satrans_head_desc* h) {
    if (!d || !h || satrans_layer_validate(d, "layer_bwd_head")) return 0;
    if (d->flags & (SATRANS_X_SORTED | SATRANS_Y_SORTED)) return 0;
    return satrans_layer_impl() == 0 && satrans_layer_fused_supported(d) && satrans_layer_bwd_head_fused_supported(d, h);
}

int64_t satrans_layer_bwd_head_scratch_floats(const satrans_layer_desc* d, int n_dense) {
    if (!d || satrans_layer_validate(d, "layer_bwd_head")) return -1;
    return satrans_layer_bwd_head_scratch_floats_fused(d, n_dense);
}

int satrans_layer_bwd_head(const satrans_layer_desc* d, const satrans_head_desc* h, float* dx, float* slabs, float* g_wq,
                           float* g_wk, float* g_wv, float* g_wo, float* g_ln, float* g_lnq, float* g_lnk, float* g_tab_q,
                           float* g_tab_k, void* stream) {
    int rc = satrans_layer_validate(d, "layer_bwd_head");
    if (rc) return rc;
    SATRANS_REQUIRE(h, SATRANS_E_BADARG, "layer_bwd_head: null head descriptor");
    SATRANS_REQUIRE(satrans_layer_bwd_head_supported(d, h), SATRANS_E_UNSUPPORTED,
                    "layer_bwd_head: not built for this layer / head (use satrans_layer_fwd + satrans_head_loss + satrans_layer_bwd)");
    return satrans_layer_bwd_head_fused(d, h, dx, slabs, g_wq, g_wk, g_wv, g_wo, g_ln, g_lnq, g_lnk, g_tab_q, g_tab_k, stream);
}

// backward kernels without their reduction launches + one reduction for all layers of a step: fused kernels only
int satrans_layer_bwd_deferred_supported(const satrans_layer_desc* d) {
    if (!d || satrans_layer_validate(d, "layer_bwd_launch")) return 0;
    return satrans_layer_impl() == 0 && satrans_layer_bwd_fused_supported(d);
}

int satrans_layer_bwd_launch(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, void* stream) {
    int rc = satrans_layer_validate(d, "layer_bwd_launch");
    if (rc) return rc;
    SATRANS_REQUIRE(satrans_layer_bwd_deferred_supported(d), SATRANS_E_UNSUPPORTED, "layer_bwd_launch: fused kernels only (use satrans_layer_bwd)");
    return satrans_layer_bwd_launch_fused(d, dy, dx, slabs, stream);
}

int satrans_layer_bwd_head_launch(const satrans_layer_desc* d, const satrans_head_desc* h, float* dx, float* slabs, void* stream) {
    int rc = satrans_layer_validate(d, "layer_bwd_head_launch");
    if (rc) return rc;
    SATRANS_REQUIRE(h && satrans_layer_bwd_head_supported(d, h), SATRANS_E_UNSUPPORTED, "layer_bwd_head_launch: not built for this layer / head");
    return satrans_layer_bwd_head_launch_fused(d, h, dx, slabs, stream);
}

int satrans_layer_bwd_reduce(int n, const satrans_layer_desc* const* h_descs, float* const* h_slabs,
                             const satrans_layer_grads* h_grads, const satrans_head_desc* head, void* stream) {
    SATRANS_REQUIRE(n >= 1 && h_descs && h_slabs && h_grads, SATRANS_E_BADARG, "layer_bwd_reduce: null argument");
    for (int l = 0; l < n; ++l) {
        int rc = satrans_layer_validate(h_descs[l], "layer_bwd_reduce");
        if (rc) return rc;
        SATRANS_REQUIRE(satrans_layer_bwd_deferred_supported(h_descs[l]), SATRANS_E_UNSUPPORTED, "layer_bwd_reduce: fused kernels only");
    }
    return satrans_layer_bwd_reduce_fused(n, h_descs, h_slabs, h_grads, head, stream);
}

}  // extern "C"
