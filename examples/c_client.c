/* A caller of libmdc.so written in plain C99: no C++, no torch, no Python -- only include/mdc.h and the HIP
 * runtime's C API for the device buffers the caller owns.  It does for a deployed net (CNN.ipynb cell 6) what the
 * reference does with Keras (cnn.py:147 load_weights, cnn.py:198 predict, cnn.py:209 argmax):
 *
 *     c_client <weights.bin> <frames.bin> <n> <out.bin> [F [lanes]]
 *
 * lanes > 1: the one-process multi-GPU form of BASELINE.json configs[3] ("per-GPU HIP streams") -- one model handle per
 * visible device, lanes streams dealt round-robin over the devices, the batch cut into contiguous shards, EVERY forward
 * enqueued from this one host thread before the first synchronisation, each stream synchronised once.
 * lanes = 0: no device buffer in the caller at all -- the host arrays go through mdc_predict_host (the library's
 * pinned-ring driver: copies, kernels and result copies overlapped on its own streams), as cnn.py:198 hands numpy arrays.
 *
 * weights.bin: float32 conv kernel (2F, HWIO), conv bias (F), dense kernel (258F x 3), dense bias (3);
 * frames.bin:  n x 2 x 128 float32;   out.bin: n x 3 float32 probabilities, then n int32 labels.
 *
 * Build (tests/test_c_client.py does exactly this):
 *     gcc -std=c99 -Wall -Werror -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_client.c \
 *         -Lmodulationdetectioncnn_amd -lmdc -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,... -o c_client
 */
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_runtime_api.h>

#include "mdc.h"

static void* read_file(const char* path, size_t bytes) {
    FILE* f = fopen(path, "rb");
    void* p = malloc(bytes ? bytes : 1);
    if (!f || !p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot read %zu bytes from %s\n", bytes, path); exit(2); }
    fclose(f);
    return p;
}

#define MDC_CHECK(call) do { int rc_ = (call); if (rc_ != MDC_OK) { \
    fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mdc_last_error()); return 1; } } while (0)
#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

#define MAX_LANES 16
#define MAX_DEVS 16

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s weights.bin frames.bin n out.bin [F [lanes]]\n", argv[0]); return 2; }
    const long n = atol(argv[3]);
    const int F = argc > 5 ? atoi(argv[5]) : 3;
    int lanes = argc > 6 ? atoi(argv[6]) : 1;
    if (lanes < 0 || lanes > MAX_LANES) { fprintf(stderr, "lanes must be 0..%d\n", MAX_LANES); return 2; }
    const int host_mode = lanes == 0;
    if (host_mode) lanes = 1;
    const size_t nk0 = 2 * (size_t)F, nb0 = (size_t)F, nk1 = 258 * (size_t)F * 3, nb1 = 3;
    float* w = (float*)read_file(argv[1], (nk0 + nb0 + nk1 + nb1) * sizeof(float));
    float* x = (float*)read_file(argv[2], (size_t)n * 256 * sizeof(float));

    if (mdc_abi_version() != MDC_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    if (ndev > MAX_DEVS) ndev = MAX_DEVS;
    if (ndev > lanes) ndev = lanes;                 /* one model per device actually used */
    mdc_topology topo = {MDC_KIND_DEPLOYED, 0, 0, 3, {0, 0, 0, 0}};
    topo.filters = F;
    mdc_model* model[MAX_DEVS];
    for (int d = 0; d < ndev; ++d) {
        model[d] = NULL;
        MDC_CHECK(mdc_create(&topo, d, &model[d]));
        size_t ke = 0, be = 0;
        MDC_CHECK(mdc_layer_sizes(model[d], 0, &ke, &be));
        if (mdc_num_layers(model[d]) != 2 || ke != nk0 || be != nb0) { fprintf(stderr, "unexpected layer sizes\n"); return 1; }
        MDC_CHECK(mdc_set_weights(model[d], 0, w, nk0, w + nk0, nb0));
        MDC_CHECK(mdc_set_weights(model[d], 1, w + nk0 + nb0, nk1, w + nk0 + nb0 + nk1, nb1));
        /* calling forward before finalize is an ordering error the library reports, it does not crash */
        if (mdc_forward(model[d], x, n, NULL, NULL, NULL, MDC_TAP_NONE, NULL, 0, NULL) != MDC_ESTATE) { fprintf(stderr, "expected MDC_ESTATE\n"); return 1; }
        MDC_CHECK(mdc_finalize(model[d], MDC_F32));
    }

    float* p = (float*)malloc((size_t)(n ? n : 1) * 3 * sizeof(float));
    int32_t* l = (int32_t*)malloc((size_t)(n ? n : 1) * sizeof(int32_t));
    float *x_dev[MAX_LANES], *p_dev[MAX_LANES];
    int32_t* l_dev[MAX_LANES];
    hipStream_t s[MAX_LANES];
    long lo[MAX_LANES + 1];
    if (host_mode) {      /* test_Y_hat = model.predict(X_test): host arrays in, host arrays out, chunks of 4,096 frames */
        MDC_CHECK(mdc_predict_host(model[0], x, n, p, l, 4096));
        lanes = 0;
    }
    for (int i = 0; i <= lanes; ++i) lo[i] = lanes ? (long)(((long long)i * n) / lanes) : 0;      /* contiguous, exhaustive shards */
    /* enqueue everything: uploads, forwards, downloads -- no synchronisation in this loop */
    for (int i = 0; i < lanes; ++i) {
        const int d = i % ndev;
        const long m = lo[i + 1] - lo[i];
        HIP_CHECK(hipSetDevice(d));
        HIP_CHECK(hipStreamCreate(&s[i]));
        HIP_CHECK(hipMalloc((void**)&x_dev[i], (size_t)(m ? m : 1) * 256 * sizeof(float)));
        HIP_CHECK(hipMalloc((void**)&p_dev[i], (size_t)(m ? m : 1) * 3 * sizeof(float)));
        HIP_CHECK(hipMalloc((void**)&l_dev[i], (size_t)(m ? m : 1) * sizeof(int32_t)));
        HIP_CHECK(hipMemcpyAsync(x_dev[i], x + lo[i] * 256, (size_t)m * 256 * sizeof(float), hipMemcpyHostToDevice, s[i]));
        MDC_CHECK(mdc_forward(model[d], x_dev[i], m, p_dev[i], l_dev[i], NULL, MDC_TAP_NONE, NULL, mdc_workspace_bytes(model[d], m), s[i]));
        HIP_CHECK(hipMemcpyAsync(p + lo[i] * 3, p_dev[i], (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost, s[i]));
        HIP_CHECK(hipMemcpyAsync(l + lo[i], l_dev[i], (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, s[i]));
    }
    for (int i = 0; i < lanes; ++i) HIP_CHECK(hipStreamSynchronize(s[i]));              /* ONE sync per lane */

    FILE* out = fopen(argv[4], "wb");
    if (!out || fwrite(p, sizeof(float), (size_t)n * 3, out) != (size_t)n * 3 || fwrite(l, sizeof(int32_t), (size_t)n, out) != (size_t)n) {
        fprintf(stderr, "cannot write %s\n", argv[4]);
        return 1;
    }
    fclose(out);
    for (int d = 0; d < ndev; ++d) mdc_destroy(model[d]);
    for (int i = 0; i < lanes; ++i) {
        HIP_CHECK(hipFree(x_dev[i])); HIP_CHECK(hipFree(p_dev[i])); HIP_CHECK(hipFree(l_dev[i]));
        HIP_CHECK(hipStreamDestroy(s[i]));
    }
    free(w); free(x); free(p); free(l);
    if (host_mode) printf("c_client: %ld frames classified from host buffers (mdc_predict_host)\n", n);
    else printf("c_client: %ld frames classified on %d device(s), %d stream(s)\n", n, ndev, lanes);
    return 0;
}
