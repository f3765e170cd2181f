// Host -> device paths for 1 GiB of pageable frames (what model.predict(numpy) hands over): plain hipMemcpy from pageable
// memory; hipHostRegister in place + one DMA; T threads copying into a pinned ring with the DMA of chunk i overlapping the
// memcpy of chunk i+1.  Build: hipcc -O3 --offload-arch=gfx950 -pthread host_path.hip -o host_path
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_copy(char* dst, const char* src, size_t bytes, int T) {
    if (T <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = (bytes / T + 4095) & ~size_t(4095);
    for (int t = 0; t < T; ++t) {
        const size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
        if (lo < hi) th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto& t : th) t.join();
}
int main() {
    const size_t bytes = (size_t)1 << 30, chunk = (size_t)64 << 20;
    char* src = (char*)malloc(bytes);
    for (size_t i = 0; i < bytes; i += 4096) src[i] = (char)i;      // touch every page
    memset(src, 1, bytes);
    char* dev; CK(hipMalloc(&dev, bytes));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 2; ++rep) {
        double t = now(); CK(hipMemcpy(dev, src, bytes, hipMemcpyHostToDevice)); double d = now() - t;
        printf("hipMemcpy pageable: %.1f ms  %.1f GB/s\n", d * 1e3, bytes / d / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t = now(); CK(hipHostRegister(src, bytes, hipHostRegisterDefault)); double r = now() - t;
        double t1 = now(); CK(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double c = now() - t1;
        double t2 = now(); CK(hipHostUnregister(src)); double u = now() - t2;
        printf("register %.1f ms + copy %.1f ms (%.1f GB/s) + unregister %.1f ms = %.1f ms  %.1f GB/s\n", r * 1e3, c * 1e3, bytes / c / 1e9, u * 1e3, (r + c + u) * 1e3, bytes / (r + c + u) / 1e9);
    }
    const int S = 3;
    char* pin[S]; hipEvent_t ev[S];
    for (int i = 0; i < S; ++i) { CK(hipHostMalloc(&pin[i], chunk, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); memset(pin[i], 0, chunk); }
    for (int T : {1, 2, 4, 8, 12}) {
        for (int rep = 0; rep < 2; ++rep) {
            double t = now();
            for (size_t off = 0, i = 0; off < bytes; off += chunk, ++i) {
                const int k = i % S;
                if (i >= (size_t)S) CK(hipEventSynchronize(ev[k]));
                par_copy(pin[k], src + off, chunk, T);
                CK(hipMemcpyAsync(dev + off, pin[k], chunk, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(ev[k], s));
            }
            CK(hipStreamSynchronize(s));
            double d = now() - t;
            if (rep) printf("pinned ring, %2d copy threads: %.1f ms  %.1f GB/s\n", T, d * 1e3, bytes / d / 1e9);
        }
    }
    {   // the DMA alone from pinned memory
        double t = now();
        for (int i = 0; i < 16; ++i) CK(hipMemcpyAsync(dev + (size_t)i * chunk, pin[i % S], chunk, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double d = now() - t;
        printf("DMA from pinned memory alone: %.1f ms  %.1f GB/s\n", d * 1e3, bytes / d / 1e9);
    }
    return 0;
}
