// placeholder (filled in below in this round): canonical VT-CNN2 (T3)
#include "mdc_internal.h"
namespace mdc {
int vtcnn2_pack(mdc_model*) { set_error("vtcnn2 kernels not built yet"); return MDC_ENOTSUP; }
size_t vtcnn2_workspace_bytes(const mdc_model*, int64_t) { return 0; }
int vtcnn2_forward(const mdc_model*, const float*, int64_t, float*, int32_t*, float*, int, void*, size_t, hipStream_t) { set_error("vtcnn2 kernels not built yet"); return MDC_ENOTSUP; }
}
