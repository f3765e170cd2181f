// Callers on either side of the forward path:
//   mdc_confusion        -- the C x C confusion counts of cnn.py:205-216 / 242-255 (conf[true][argmax] += 1) on the
//                           device, so an evaluation over a sharded batch exchanges C*C integers, not N labels;
//   mdc_confusion_binned -- the same per SNR bin (the loop of cnn.py:228-259) in ONE launch: a B x C x C histogram;
//   mdc_crossentropy     -- the sum behind `score = model.evaluate(X_test, Y_test)` (cnn.py:153: the model is compiled with
//                           loss='categorical_crossentropy' and no metric, so the score IS the mean loss);
//   mdc_iq_u8_to_frames / mdc_iq_u8_windows
//                        -- raw SDR samples (unsigned 8-bit interleaved I,Q, the format of the RTL-SDR front-end the
//                           reference's README.md:5 describes) -> (n,2,128) f32 frames.
#include "mdc_internal.h"

namespace mdc {

namespace {

constexpr int kMaxConfClasses = 32;
constexpr int kMaxLdsCells = 12288;       // 48 KiB of LDS histogram; larger (bins x C x C) tables count straight into global memory

// counts[(b*C + t)*C + p] += #{i : bin[i] == b, truth[i] == t, pred[i] == p}  (bin == NULL: one bin).  Entries with a
// label outside [0,C) or a bin outside [0,B) are counted in *bad (if given).  LDS = true: per-work-group histogram in
// LDS, one global atomic per non-zero cell at the end.
template <bool LDS>
__global__ __launch_bounds__(256) void confusion_kernel(const int* __restrict__ truth, const int* __restrict__ pred, const int* __restrict__ bin,
                                                        long n, int C, int B, unsigned long long* __restrict__ counts,
                                                        unsigned long long* __restrict__ bad) {
    extern __shared__ unsigned hist[];
    const int cells = B * C * C;
    if (LDS) {
        for (int i = threadIdx.x; i <= cells; i += blockDim.x) hist[i] = 0u;
        __syncthreads();
    }
    unsigned nbad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int t = truth[i], p = pred[i], b = bin ? bin[i] : 0;
        const bool ok = (unsigned)t < (unsigned)C && (unsigned)p < (unsigned)C && (unsigned)b < (unsigned)B;
        if (LDS) atomicAdd(&hist[ok ? (b * C + t) * C + p : cells], 1u);
        else if (ok) atomicAdd(&counts[(b * C + t) * C + p], 1ull);
        else ++nbad;
    }
    if (LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < cells; i += blockDim.x)
            if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
        if (threadIdx.x == 0 && bad && hist[cells]) atomicAdd(bad, (unsigned long long)hist[cells]);
    } else if (bad && nbad) {
        atomicAdd(bad, (unsigned long long)nbad);
    }
}

// Keras' categorical_crossentropy on PROBABILITIES (the reference's model ends in Activation('softmax') + Reshape, so the loss
// sees the softmax output, not logits): row scaled to sum 1, clipped to [1e-7, 1 - 1e-7], -log of the true class's entry.
// f64 accumulation: per-thread partial, LDS tree per work-group, one atomic per work-group.
__global__ __launch_bounds__(256) void crossentropy_kernel(const float* __restrict__ probs, const int* __restrict__ truth, long n, int C,
                                                           double* __restrict__ loss_sum, unsigned long long* __restrict__ bad) {
    __shared__ double part[256];
    double acc = 0.0;
    unsigned nbad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int t = truth[i];
        if ((unsigned)t >= (unsigned)C) { ++nbad; continue; }
        const float* row = probs + i * C;
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += row[c];
        float p = row[t] / sum;
        // a NaN row (Inf / NaN samples in, or a row summing to 0) stays NaN, as Keras' clip_by_value leaves it (fmaxf alone
        // would turn it into 1e-7 and the score into a finite number)
        if (p == p) p = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
        acc -= (double)logf(p);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && part[0] != 0.0) atomicAdd(loss_sum, part[0]);
    if (bad && nbad) atomicAdd(bad, (unsigned long long)nbad);
}

// one thread = 4 consecutive bytes of a window = samples (I[2k], Q[2k], I[2k+1], Q[2k+1]); writes two floats to each
// row.  Window f starts at byte 2*hop*f of the capture (hop = 128: disjoint frames); an odd hop leaves the 4 bytes only
// 2-byte aligned, so they are fetched as two 16-bit loads then.
template <bool ODD>
__global__ __launch_bounds__(256) void iq_u8_kernel(const unsigned char* __restrict__ iq, long n, long hop, float scale, float* __restrict__ x) {
    const long total = n * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long f = i >> 6;
        const int k = (int)(i & 63);
        const unsigned char* src = iq + 2 * hop * f + 4 * k;
        unsigned w;
        if (ODD) w = (unsigned)reinterpret_cast<const unsigned short*>(src)[0] | ((unsigned)reinterpret_cast<const unsigned short*>(src)[1] << 16);
        else w = *reinterpret_cast<const unsigned*>(src);
        float* row_i = x + f * kFrameFloats + 2 * k;
        *reinterpret_cast<float2*>(row_i) = make_float2(((float)(w & 0xFFu) - 127.5f) * scale, ((float)((w >> 16) & 0xFFu) - 127.5f) * scale);
        *reinterpret_cast<float2*>(row_i + kSamples) = make_float2(((float)((w >> 8) & 0xFFu) - 127.5f) * scale, ((float)(w >> 24) - 127.5f) * scale);
    }
}

}  // namespace

int confusion_launch(const int32_t* truth, const int32_t* pred, const int32_t* bin, int64_t n, int classes, int bins, int64_t* counts,
                     int64_t* bad, hipStream_t s) {
    if (classes < 1 || classes > kMaxConfClasses) { set_error("mdc_confusion: classes must be 1..%d (got %d)", kMaxConfClasses, classes); return MDC_EINVAL; }
    if (bins < 1 || bins > 65536) { set_error("mdc_confusion: bins must be 1..65536 (got %d)", bins); return MDC_EINVAL; }
    if (n == 0) return MDC_OK;
    long grid = (n + 255) / 256;
    if (grid > 1024) grid = 1024;
    const long cells = (long)bins * classes * classes;
    auto* c = reinterpret_cast<unsigned long long*>(counts);
    auto* b = reinterpret_cast<unsigned long long*>(bad);
    if (cells <= kMaxLdsCells)
        hipLaunchKernelGGL(confusion_kernel<true>, dim3((unsigned)grid), dim3(256), (size_t)(cells + 1) * sizeof(unsigned), s, truth, pred, bin, (long)n,
                           classes, bins, c, b);
    else
        hipLaunchKernelGGL(confusion_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, truth, pred, bin, (long)n, classes, bins, c, b);
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

int crossentropy_launch(const float* probs, const int32_t* truth, int64_t n, int classes, double* loss_sum, int64_t* bad, hipStream_t s) {
    if (classes < 1 || classes > kMaxConfClasses) { set_error("mdc_crossentropy: classes must be 1..%d (got %d)", kMaxConfClasses, classes); return MDC_EINVAL; }
    if (n == 0) return MDC_OK;
    long grid = (n + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(crossentropy_kernel, dim3((unsigned)grid), dim3(256), 0, s, probs, truth, (long)n, classes, loss_sum,
                       reinterpret_cast<unsigned long long*>(bad));
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

int iq_u8_launch(const uint8_t* iq, int64_t n, int64_t hop, float scale, float* x, hipStream_t s) {
    if (n == 0) return MDC_OK;
    long grid = (n * 64 + 255) / 256;
    if (grid > 8192) grid = 8192;
    const bool odd = (hop & 1) != 0 || (reinterpret_cast<uintptr_t>(iq) & 3) != 0;
    if (odd) hipLaunchKernelGGL(iq_u8_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, iq, (long)n, (long)hop, scale, x);
    else     hipLaunchKernelGGL(iq_u8_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, iq, (long)n, (long)hop, scale, x);
    MDC_HIP(hipGetLastError());
    return MDC_OK;
}

}  // namespace mdc
