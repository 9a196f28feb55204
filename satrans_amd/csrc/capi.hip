// Error reporting and the shape dispatch of the public layer entry points.
#include <stdarg.h>

#include "common.h"

namespace satrans {
static thread_local char g_error[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
}  // namespace satrans

extern "C" {

int satrans_layer_fwd_lds(const satrans_layer_desc* d, float* y, float* att, void* stream);
int64_t satrans_layer_bwd_slab_floats_lds(const satrans_layer_desc* d);
int satrans_layer_bwd_lds(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, float* g_wq, float* g_wk,
                          float* g_wv, float* g_wo, float* g_ln, float* g_lnq, float* g_lnk, float* g_tab_q,
                          float* g_tab_k, void* stream);

const char* satrans_last_error(void) { return satrans::g_error; }
int satrans_abi_version(void) { return SATRANS_ABI_VERSION; }

int satrans_layer_fwd(const satrans_layer_desc* d, float* y, float* att, void* stream) {
    return satrans_layer_fwd_lds(d, y, att, stream);
}

int64_t satrans_layer_bwd_slab_floats(const satrans_layer_desc* d) { return satrans_layer_bwd_slab_floats_lds(d); }

int satrans_layer_bwd(const satrans_layer_desc* d, const float* dy, float* dx, float* slabs, float* g_wq, float* g_wk,
                      float* g_wv, float* g_wo, float* g_ln, float* g_lnq, float* g_lnk, float* g_tab_q, float* g_tab_k,
                      void* stream) {
    return satrans_layer_bwd_lds(d, dy, dx, slabs, g_wq, g_wk, g_wv, g_wo, g_ln, g_lnq, g_lnk, g_tab_q, g_tab_k, stream);
}

}  // extern "C"
