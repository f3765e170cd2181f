// model.predict(X) on HOST buffers (cnn.py:198, 237: X_test is a numpy array in host memory; the FFI of a host-language
// caller hands over exactly that).  mdc_forward wants its frames in HBM; this is the native driver in front of it:
//
//   caller's pageable memory --(copy threads)--> pinned ring --(DMA, copy stream)--> HBM slot
//        --(mdc_forward / mdc_forward_iq_u8, compute stream)--> results slot --(DMA, results stream)--> pinned --> caller
//
// Three slots of `chunk` frames (the first two chunks of a large batch are shorter: the fill ramp in run_pipeline): while chunk i is computed, chunk i+1 crosses PCIe and chunk i+2 is being copied into
// pinned memory by the host threads, so the call runs at max(PCIe, kernel) instead of their sum (1 GiB of frames:
// 18.9 ms of DMA at 57 GB/s; a plain hipMemcpy from pageable memory takes 63 ms the first time it sees the buffer;
// tools/microbench/host_path.hip).  Memory that is already pinned (hipHostMalloc / hipHostRegister, e.g. a pinned
// torch tensor) is DMA'd from where it lies.  Results are bit-identical to the device entry points: the same kernels
// run on the same frames, chunking does not change a frame's arithmetic.
//
// The context (streams, events, pinned and device slots, workspace) belongs to the model, is created on first use,
// grows on demand and is freed by mdc_destroy.  Calls on one model are serialised by a mutex (the call is synchronous:
// it returns when the caller's output buffers are complete, like Keras' predict).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <thread>

#include "mdc_internal.h"

namespace mdc {
namespace {

constexpr int kSlots = 3;
constexpr int64_t kDefaultChunkFrames = 65536;      // 64 MiB of f32 frames per slot; one T3 launch chunk
constexpr int kCopyThreads = 4;                     // 2 reach 50 GB/s, 4 the DMA's 54-57 GB/s on the MI355X host

// A few helper threads that live for one call and split each staging memcpy with the calling thread.
class CopyPool {
public:
    // A helper that cannot be started (EAGAIN under a thread limit: std::system_error) is simply not there: the pool runs
    // with the helpers it got -- or none, the caller copies alone -- and never unwinds with joinable threads in th_
    // (a half-built pool's destructor does not run: std::terminate, which the ABI's catch-all cannot stop).
    explicit CopyPool(int helpers) {
        try {
            th_.reserve(helpers > 0 ? (size_t)helpers : 0);
            for (int i = 0; i < helpers; ++i) th_.emplace_back([this, i] { run(i); });
        } catch (...) {
        }
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread& t : th_) t.join();
    }
    void copy(char* dst, const char* src, size_t bytes) {
        const int parts = (int)th_.size() + 1;
        if (parts == 1 || bytes < ((size_t)1 << 20)) { std::memcpy(dst, src, bytes); return; }
        const size_t per = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> g(mu_);
            dst_ = dst; src_ = src; bytes_ = bytes; per_ = per;
            pending_ = (int)th_.size();
            ++gen_;
        }
        cv_.notify_all();
        part(parts - 1);      // the caller takes the last part
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }

private:
    void part(int i) {
        const size_t lo = (size_t)i * per_;
        if (lo < bytes_) std::memcpy(dst_ + lo, src_ + lo, std::min(per_, bytes_ - lo));
    }
    void run(int i) {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            part(i);
            {
                std::lock_guard<std::mutex> g(mu_);
                --pending_;
            }
            done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t bytes_ = 0, per_ = 0;
    int pending_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};

struct Slot {
    char* pin_in = nullptr;
    char* d_in = nullptr;
    float* d_probs = nullptr;
    int32_t* d_labels = nullptr;
    char* pin_out = nullptr;      // probabilities, then labels
    hipEvent_t in_done = nullptr, comp_done = nullptr, out_done = nullptr;
    int64_t start = 0, count = 0;
    bool busy = false;
};

}  // namespace

struct HostCtx {
    hipStream_t copy_s = nullptr, comp_s = nullptr, out_s = nullptr;
    Slot slot[kSlots];
    size_t in_cap = 0;            // bytes of input per slot
    int64_t out_cap = 0;          // frames of output per slot
    void* ws = nullptr;
    size_t ws_bytes = 0;
};

void host_ctx_free(mdc_model* m) {
    HostCtx* c = static_cast<HostCtx*>(m->host_ctx);
    if (!c) return;
    for (Slot& s : c->slot) {
        if (s.pin_in) (void)hipHostFree(s.pin_in);
        if (s.pin_out) (void)hipHostFree(s.pin_out);
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_probs) (void)hipFree(s.d_probs);
        if (s.d_labels) (void)hipFree(s.d_labels);
        if (s.in_done) (void)hipEventDestroy(s.in_done);
        if (s.comp_done) (void)hipEventDestroy(s.comp_done);
        if (s.out_done) (void)hipEventDestroy(s.out_done);
    }
    if (c->ws) (void)hipFree(c->ws);
    if (c->copy_s) (void)hipStreamDestroy(c->copy_s);
    if (c->comp_s) (void)hipStreamDestroy(c->comp_s);
    if (c->out_s) (void)hipStreamDestroy(c->out_s);
    delete c;
    m->host_ctx = nullptr;
}

namespace {

int ctx_prepare(mdc_model* m, size_t in_bytes, int64_t frames) {
    HostCtx* c = static_cast<HostCtx*>(m->host_ctx);
    if (!c) {
        c = new HostCtx();
        m->host_ctx = c;
        auto create = [&]() -> int {
            MDC_HIP(hipStreamCreateWithFlags(&c->copy_s, hipStreamNonBlocking));
            MDC_HIP(hipStreamCreateWithFlags(&c->comp_s, hipStreamNonBlocking));
            MDC_HIP(hipStreamCreateWithFlags(&c->out_s, hipStreamNonBlocking));
            for (Slot& s : c->slot) {
                MDC_HIP(hipEventCreateWithFlags(&s.in_done, hipEventDisableTiming));
                MDC_HIP(hipEventCreateWithFlags(&s.comp_done, hipEventDisableTiming));
                MDC_HIP(hipEventCreateWithFlags(&s.out_done, hipEventDisableTiming));
            }
            return MDC_OK;
        };
        const int rc = create();
        if (rc != MDC_OK) { host_ctx_free(m); return rc; }      // never keep a half-made context for the next call
    }
    const int C = m->topo.classes;
    if (in_bytes > c->in_cap) {
        for (Slot& s : c->slot) {
            if (s.pin_in) { (void)hipHostFree(s.pin_in); s.pin_in = nullptr; }
            if (s.d_in) { (void)hipFree(s.d_in); s.d_in = nullptr; }
        }
        c->in_cap = 0;
        for (Slot& s : c->slot) {
            if (hipHostMalloc(reinterpret_cast<void**>(&s.pin_in), in_bytes, hipHostMallocDefault) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&s.d_in), in_bytes) != hipSuccess) {
                (void)hipGetLastError();
                set_error("host path: cannot allocate %zu bytes of pinned / device staging per slot", in_bytes);
                return MDC_ENOMEM;
            }
        }
        c->in_cap = in_bytes;
    }
    if (frames > c->out_cap) {
        for (Slot& s : c->slot) {
            if (s.pin_out) { (void)hipHostFree(s.pin_out); s.pin_out = nullptr; }
            if (s.d_probs) { (void)hipFree(s.d_probs); s.d_probs = nullptr; }
            if (s.d_labels) { (void)hipFree(s.d_labels); s.d_labels = nullptr; }
        }
        c->out_cap = 0;
        for (Slot& s : c->slot) {
            if (hipHostMalloc(reinterpret_cast<void**>(&s.pin_out), (size_t)frames * (C + 1) * 4, hipHostMallocDefault) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&s.d_probs), (size_t)frames * C * 4) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&s.d_labels), (size_t)frames * 4) != hipSuccess) {
                (void)hipGetLastError();
                set_error("host path: cannot allocate result staging for %lld frames per slot", (long long)frames);
                return MDC_ENOMEM;
            }
        }
        c->out_cap = frames;
    }
    const size_t need = mdc_workspace_bytes(m, frames);
    if (need > c->ws_bytes) {
        if (c->ws) { (void)hipFree(c->ws); c->ws = nullptr; c->ws_bytes = 0; }
        if (hipMalloc(&c->ws, need) != hipSuccess) {
            (void)hipGetLastError();
            set_error("host path: cannot allocate a %zu-byte workspace", need);
            return MDC_ENOMEM;
        }
        c->ws_bytes = need;
    }
    return MDC_OK;
}

// (the first byte decides: a caller that pins its buffer pins all of it)
// 0 = pageable host memory (or unknown to HIP), 1 = pinned host memory, -1 = DEVICE memory: not a host buffer at all
int host_kind(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (a.type == hipMemoryTypeDevice) return -1;
    return a.type == hipMemoryTypeHost ? 1 : 0;
}
bool is_pinned(const void* p) { return host_kind(p) == 1; }

// in_range(start, count) -> (byte offset, byte count) of the input those windows read; launch(d_in, count, slot, ctx)
template <class InRange, class Launch>
int run_pipeline(mdc_model* m, const char* src, int64_t n, int64_t chunk, bool ramp, float* probs_host, int32_t* labels_host,
                 InRange in_range, Launch launch) {
    HostCtx* c = static_cast<HostCtx*>(m->host_ctx);
    const int C = m->topo.classes;
    const bool direct = is_pinned(src);
    // helper threads only where they pay: a few-frame call (one window of a live stream) must not spend 100 us starting them
    size_t total_off = 0, total_bytes = 0;
    in_range(0, n, &total_off, &total_bytes);
    const bool helpers = !direct && total_bytes >= ((size_t)8 << 20);
    CopyPool pool(helpers ? std::max(0, std::min<int>(kCopyThreads, (int)std::thread::hardware_concurrency()) - 1) : 0);
    auto retire = [&](Slot& s) -> int {
        if (!s.busy) return MDC_OK;
        MDC_HIP(hipEventSynchronize(s.out_done));
        if (probs_host) std::memcpy(probs_host + s.start * C, s.pin_out, (size_t)s.count * C * 4);
        if (labels_host) std::memcpy(labels_host + s.start, s.pin_out + (size_t)c->out_cap * C * 4, (size_t)s.count * 4);
        s.busy = false;
        return MDC_OK;
    };
    int rc = MDC_OK;
    int64_t i = 0;
    // ramp (the library's own chunking of a large batch only): the first two chunks are a quarter and a half slot long, so
    // the first kernel starts after 16 MiB have been staged and copied instead of 64 (the pipeline's fill time is paid
    // once per call: 2 of 33.5 ms for 2^20 frames of VT-CNN2 bf16); results do not depend on chunking
    int64_t count = 0;
    for (int64_t start = 0; start < n && rc == MDC_OK; start += count, ++i) {
        Slot& s = c->slot[i % kSlots];
        if ((rc = retire(s)) != MDC_OK) break;      // chunk i - 3 is done with the slot's buffers
        // (never longer than a slot: with a large hop a slot holds fewer than 256 windows, and then there is no ramp)
        const int64_t len = ramp && i < 2 ? std::min(chunk, std::max<int64_t>(256, (chunk >> (2 - i)) & ~(int64_t)255)) : chunk;
        count = std::min(len, n - start);
        size_t off = 0, bytes = 0;
        in_range(start, count, &off, &bytes);
        if (count > c->out_cap || bytes > c->in_cap) {      // the slots were sized for `chunk`: refuse before any copy
            set_error("host path: a chunk of %lld frames (%zu input bytes) exceeds its slot (%lld frames, %zu bytes)",
                      (long long)count, bytes, (long long)c->out_cap, c->in_cap);
            rc = MDC_EINVAL;
            break;
        }
        hipError_t e;
        if (direct) {
            e = hipMemcpyAsync(s.d_in, src + off, bytes, hipMemcpyHostToDevice, c->copy_s);
        } else {
            pool.copy(s.pin_in, src + off, bytes);
            e = hipMemcpyAsync(s.d_in, s.pin_in, bytes, hipMemcpyHostToDevice, c->copy_s);
        }
        if (e == hipSuccess) e = hipEventRecord(s.in_done, c->copy_s);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->comp_s, s.in_done, 0);
        if (e != hipSuccess) { set_error("host path: input copy failed: %s", hipGetErrorString(e)); rc = MDC_EIO; break; }
        if ((rc = launch(s, count, c)) != MDC_OK) break;
        e = hipEventRecord(s.comp_done, c->comp_s);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->out_s, s.comp_done, 0);
        if (e == hipSuccess && probs_host) e = hipMemcpyAsync(s.pin_out, s.d_probs, (size_t)count * C * 4, hipMemcpyDeviceToHost, c->out_s);
        if (e == hipSuccess && labels_host) e = hipMemcpyAsync(s.pin_out + (size_t)c->out_cap * C * 4, s.d_labels, (size_t)count * 4, hipMemcpyDeviceToHost, c->out_s);
        if (e == hipSuccess) e = hipEventRecord(s.out_done, c->out_s);
        if (e != hipSuccess) { set_error("host path: result copy failed: %s", hipGetErrorString(e)); rc = MDC_EIO; break; }
        s.start = start;
        s.count = count;
        s.busy = true;
    }
    // drain in chunk order: the oldest busy slot is the one after the last one used
    for (int k = 0; k < kSlots; ++k) {
        Slot& s = c->slot[(i + k) % kSlots];
        if (rc == MDC_OK) rc = retire(s);
        else s.busy = false;
    }
    if (rc != MDC_OK) {      // nothing may still be reading the caller's memory or writing the slots when we return
        (void)hipStreamSynchronize(c->copy_s);
        (void)hipStreamSynchronize(c->comp_s);
        (void)hipStreamSynchronize(c->out_s);
    }
    return rc;
}

// Slot length when the caller names none: 65,536 frames for large batches; a mid-size batch is cut into about four chunks
// (never below 8,192 frames: the VT-CNN2 kernels lose efficiency on smaller launches) so that its copies and kernels overlap too
int64_t default_chunk(int64_t n) {
    if (n >= 4 * kDefaultChunkFrames) return kDefaultChunkFrames;
    const int64_t q = ((n + 3) / 4 + 255) & ~(int64_t)255;
    return std::min<int64_t>(kDefaultChunkFrames, std::max<int64_t>(8192, q));
}

int check_common(const char* who, const mdc_model* m, const void* in, int64_t n, int64_t chunk) {
    if (!m) { set_error("%s: null model", who); return MDC_EINVAL; }
    if (!m->finalized) { set_error("%s: model not finalized", who); return MDC_ESTATE; }
    if (n < 0) { set_error("%s: negative frame count", who); return MDC_EINVAL; }
    if (chunk < 0) { set_error("%s: negative chunk size", who); return MDC_EINVAL; }
    if (n > 0 && !in) { set_error("%s: null input", who); return MDC_EINVAL; }
    // a device pointer here would be handed to memcpy by the staging threads: these entry points take HOST buffers
    // (frames already in HBM go to mdc_forward / mdc_forward_iq_u8)
    if (n > 0 && host_kind(in) < 0) { set_error("%s: the input lies in device memory; use mdc_forward / mdc_forward_iq_u8 for it", who); return MDC_EINVAL; }
    return MDC_OK;
}

}  // namespace

int predict_host(mdc_model* m, const float* x_host, int64_t n, float* probs_host, int32_t* labels_host, int64_t chunk_frames) {
    int rc = check_common("mdc_predict_host", m, x_host, n, chunk_frames);
    if (rc != MDC_OK || n == 0) return rc;
    std::lock_guard<std::mutex> g(m->host_mu);
    DeviceScope dev(m->device);
    if (!dev.ok) { set_error("mdc_predict_host: cannot select device %d", m->device); return MDC_EIO; }
    const int64_t chunk = std::min<int64_t>(chunk_frames > 0 ? chunk_frames : default_chunk(n), n);
    if ((rc = ctx_prepare(m, (size_t)chunk * kFrameFloats * 4, chunk)) != MDC_OK) return rc;
    return run_pipeline(
        m, reinterpret_cast<const char*>(x_host), n, chunk, chunk_frames <= 0 && n >= 4 * chunk, probs_host, labels_host,
        [](int64_t start, int64_t count, size_t* off, size_t* bytes) {
            *off = (size_t)start * kFrameFloats * 4;
            *bytes = (size_t)count * kFrameFloats * 4;
        },
        [m](Slot& s, int64_t count, HostCtx* c) {
            return mdc_forward(m, s.d_in, count, s.d_probs, s.d_labels, nullptr, MDC_TAP_NONE, c->ws, c->ws_bytes, c->comp_s);
        });
}

int predict_host_iq_u8(mdc_model* m, const uint8_t* iq_host, int64_t n, int64_t hop, float scale, float* probs_host,
                       int32_t* labels_host, int64_t chunk_frames) {
    int rc = check_common("mdc_predict_host_iq_u8", m, iq_host, n, chunk_frames);
    if (rc != MDC_OK) return rc;
    if (m->topo.kind == MDC_KIND_CNNPY) { set_error("mdc_predict_host_iq_u8: raw-IQ input exists for the deployed and vtcnn2 families"); return MDC_ENOTSUP; }
    if (hop < 1 || hop > (int64_t)1 << 24) { set_error("mdc_predict_host_iq_u8: hop must be in 1..2^24 sample pairs (got %lld)", (long long)hop); return MDC_EINVAL; }
    if (n == 0) return MDC_OK;
    std::lock_guard<std::mutex> g(m->host_mu);
    DeviceScope dev(m->device);
    if (!dev.ok) { set_error("mdc_predict_host_iq_u8: cannot select device %d", m->device); return MDC_EIO; }
    int64_t chunk = std::min<int64_t>(chunk_frames > 0 ? chunk_frames : default_chunk(n), n);
    if (chunk_frames <= 0) {      // default: a slot's input stays within the f32 path's 64 MiB however large the hop
        const int64_t fit = ((kDefaultChunkFrames * kFrameFloats * 4) - 256) / (2 * hop) + 1;
        chunk = std::max<int64_t>(1, std::min(chunk, fit));
    }
    const size_t in_bytes = (size_t)(2 * hop) * (size_t)(chunk - 1) + 256;
    if ((rc = ctx_prepare(m, in_bytes, chunk)) != MDC_OK) return rc;
    return run_pipeline(
        m, reinterpret_cast<const char*>(iq_host), n, chunk, chunk_frames <= 0 && n >= 4 * chunk, probs_host, labels_host,
        [hop](int64_t start, int64_t count, size_t* off, size_t* bytes) {
            *off = (size_t)(2 * hop) * (size_t)start;
            *bytes = (size_t)(2 * hop) * (size_t)(count - 1) + 256;
        },
        [m, hop, scale](Slot& s, int64_t count, HostCtx* c) {
            return mdc_forward_iq_u8(m, reinterpret_cast<const uint8_t*>(s.d_in), count, hop, scale, s.d_probs, s.d_labels, c->ws, c->ws_bytes,
                                     c->comp_s);
        });
}

}  // namespace mdc
