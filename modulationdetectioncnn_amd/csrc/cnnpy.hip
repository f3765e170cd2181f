// placeholder (filled in below in this round): cnn.py literal model (T4)
#include "mdc_internal.h"
namespace mdc {
int cnnpy_pack(mdc_model*) { set_error("cnnpy kernels not built yet"); return MDC_ENOTSUP; }
int cnnpy_forward(const mdc_model*, const float*, int64_t, float*, int32_t*, float*, int, hipStream_t) { set_error("cnnpy kernels not built yet"); return MDC_ENOTSUP; }
}
