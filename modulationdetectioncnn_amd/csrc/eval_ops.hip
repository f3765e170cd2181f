// Callers on either side of the forward path:
//   mdc_confusion        -- the C x C confusion counts of cnn.py:205-216 / 242-255 (conf[true][argmax] += 1) on the
//                           device, so an evaluation over a sharded batch exchanges C*C integers, not N labels;
//   mdc_iq_u8_to_frames  -- raw SDR samples (unsigned 8-bit interleaved I,Q, the format of the RTL-SDR front-end the
//                           reference's README.md:5 describes) -> (n,2,128) f32 frames.
#include "mdc_internal.h"

namespace mdc {

namespace {

constexpr int kMaxConfClasses = 32;

// counts[t*C + p] += #{i : truth[i] == t, pred[i] == p}.  Labels outside [0,C) are counted in *bad (if given).
__global__ __launch_bounds__(256) void confusion_kernel(const int* __restrict__ truth, const int* __restrict__ pred, long n, int C,
                                                        unsigned long long* __restrict__ counts, unsigned long long* __restrict__ bad) {
    __shared__ unsigned hist[kMaxConfClasses * kMaxConfClasses + 1];
    for (int i = threadIdx.x; i <= C * C; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int t = truth[i], p = pred[i];
        const bool ok = (unsigned)t < (unsigned)C && (unsigned)p < (unsigned)C;
        atomicAdd(&hist[ok ? t * C + p : C * C], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x)
        if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
    if (threadIdx.x == 0 && bad && hist[C * C]) atomicAdd(bad, (unsigned long long)hist[C * C]);
}

// one thread = 4 consecutive bytes of a frame = samples (I[2k], Q[2k], I[2k+1], Q[2k+1]); writes two floats to each row
__global__ __launch_bounds__(256) void iq_u8_kernel(const uchar4* __restrict__ iq, long n, float scale, float* __restrict__ x) {
    const long total = n * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const uchar4 b = iq[i];
        const long f = i >> 6;
        const int k = (int)(i & 63);
        float* row_i = x + f * kFrameFloats + 2 * k;
        *reinterpret_cast<float2*>(row_i) = make_float2(((float)b.x - 127.5f) * scale, ((float)b.z - 127.5f) * scale);
        *reinterpret_cast<float2*>(row_i + kSamples) = make_float2(((float)b.y - 127.5f) * scale, ((float)b.w - 127.5f) * scale);
    }
}

}  // namespace

int confusion_launch(const int32_t* truth, const int32_t* pred, int64_t n, int classes, int64_t* counts, int64_t* bad, hipStream_t s) {
    if (classes < 1 || classes > kMaxConfClasses) { set_error("mdc_confusion: classes must be 1..%d (got %d)", kMaxConfClasses, classes); return MDC_EINVAL; }
    if (n == 0) return MDC_OK;
    long grid = (n + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)grid), dim3(256), 0, s, truth, pred, (long)n, classes,
                       reinterpret_cast<unsigned long long*>(counts), reinterpret_cast<unsigned long long*>(bad));
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

int iq_u8_launch(const uint8_t* iq, int64_t n, float scale, float* x, hipStream_t s) {
    if (n == 0) return MDC_OK;
    long grid = (n * 64 + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(iq_u8_kernel, dim3((unsigned)grid), dim3(256), 0, s, reinterpret_cast<const uchar4*>(iq), (long)n, scale, x);
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
