// placeholder, replaced below in this round
#include "mdc_internal.h"
namespace mdc {
int vtcnn2_bf16_pack(mdc_model*) { set_error("bf16 kernels not built yet"); return MDC_ENOTSUP; }
int vtcnn2_bf16_conv(const mdc_model*, const float*, int64_t, void*, hipStream_t) { return MDC_ENOTSUP; }
int vtcnn2_bf16_dense1(const mdc_model*, const void*, int64_t, float*, hipStream_t) { return MDC_ENOTSUP; }
}
