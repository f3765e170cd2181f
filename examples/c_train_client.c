/* The training half of the reference (cnn.py:113, 122-147) from plain C99: no C++, no torch, no Python -- only include/mdc.h
 * and the HIP runtime's C API for the device buffers the caller owns.
 *
 *     c_train_client <weights.bin> <frames.bin> <targets.bin> <n> <n_val> <epochs> <batch> <out.bin> [F]
 *
 * weights.bin: float32 conv kernel (2F, HWIO), conv bias (F), dense kernel (258F x 3), dense bias (3) -- the initial weights
 *              (cnn.py:108-111 draws them; here the caller brings them);
 * frames.bin:  (n + n_val) x 2 x 128 float32, the first n are trained on, the last n_val validated on;
 * targets.bin: (n + n_val) x 3 float32 one-hot rows (cnn.py:74-82);
 * out.bin:     per epoch two doubles (loss, val_loss); then the BEST epoch's weights in the layout of weights.bin.
 *
 * The loop is the caller's (model.fit's epochs, ModelCheckpoint(save_best_only), EarlyStopping(patience = 5)); every
 * mini-batch is one mdc_train_batch (two launches, no synchronisation), an epoch ends with ONE mdc_trainer_read.  The
 * shuffle is an index array uploaded once per epoch (here: a fixed multiplicative permutation, so that the Python test can
 * replay it in the oracle); frames never move.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

int main(int argc, char** argv) {
    if (argc < 9) { fprintf(stderr, "usage: %s weights.bin frames.bin targets.bin n n_val epochs batch out.bin [F]\n", argv[0]); return 2; }
    const long n = atol(argv[4]), nv = atol(argv[5]);
    const int epochs = atoi(argv[6]);
    const long batch = atol(argv[7]);
    const int F = argc > 9 ? atoi(argv[9]) : 3;
    const int patience = 5;
    const size_t nk[2] = {2 * (size_t)F, 258 * (size_t)F * 3}, nb[2] = {(size_t)F, 3};
    const size_t nw = nk[0] + nb[0] + nk[1] + nb[1];
    float* w = (float*)read_file(argv[1], nw * sizeof(float));
    float* x = (float*)read_file(argv[2], (size_t)(n + nv) * 256 * sizeof(float));
    float* y = (float*)read_file(argv[3], (size_t)(n + nv) * 3 * sizeof(float));
    if (mdc_abi_version() != MDC_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    hipStream_t s;
    HIP_CHECK(hipSetDevice(0));
    HIP_CHECK(hipStreamCreate(&s));
    float *x_dev, *y_dev;
    int32_t* order_dev;
    HIP_CHECK(hipMalloc((void**)&x_dev, (size_t)(n + nv) * 256 * sizeof(float)));
    HIP_CHECK(hipMalloc((void**)&y_dev, (size_t)(n + nv) * 3 * sizeof(float)));
    HIP_CHECK(hipMalloc((void**)&order_dev, (size_t)(n > 0 ? n : 1) * sizeof(int32_t)));
    HIP_CHECK(hipMemcpy(x_dev, x, (size_t)(n + nv) * 256 * sizeof(float), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(y_dev, y, (size_t)(n + nv) * 3 * sizeof(float), hipMemcpyHostToDevice));

    /* model.compile(loss='categorical_crossentropy', optimizer='adam')                         cnn.py:113 */
    mdc_topology topo = {MDC_KIND_DEPLOYED, 0, 0, 3, {0, 0, 0, 0}};
    topo.filters = F;
    mdc_trainer* t = NULL;
    MDC_CHECK(mdc_trainer_create(&topo, 0, &t));
    if (mdc_trainer_num_layers(t) != 2) { fprintf(stderr, "unexpected layer count\n"); return 1; }
    const float* wp[2][2] = {{w, w + nk[0]}, {w + nk[0] + nb[0], w + nk[0] + nb[0] + nk[1]}};
    for (int l = 0; l < 2; ++l) {
        size_t ek = 0, eb = 0;
        MDC_CHECK(mdc_trainer_layer_sizes(t, l, &ek, &eb));
        if (ek != nk[l] || eb != nb[l]) { fprintf(stderr, "layer %d size mismatch\n", l); return 1; }
        MDC_CHECK(mdc_trainer_set_tensor(t, MDC_TRAIN_WEIGHTS, l, wp[l][0], nk[l], wp[l][1], nb[l], s));
    }

    FILE* out = fopen(argv[8], "wb");
    if (!out) { fprintf(stderr, "cannot write %s\n", argv[8]); return 2; }
    int32_t* order = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    float* best_w = (float*)malloc(nw * sizeof(float));
    memcpy(best_w, w, nw * sizeof(float));
    double best = 1e300;
    int wait = 0, ran = 0;
    for (int ep = 0; ep < epochs && wait < patience; ++ep, ++ran) {
        /* the epoch's shuffle: i -> (a i + ep) mod n with a coprime to n (the test replays it) */
        for (long i = 0; i < n; ++i) order[i] = (int32_t)((7919L * i + 13L * ep) % n);
        HIP_CHECK(hipMemcpyAsync(order_dev, order, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
        for (long first = 0; first < n; first += batch)                                     /* model.fit: one step per batch   cnn.py:135 */
            MDC_CHECK(mdc_train_batch(t, x_dev, y_dev, n, order_dev, first, n - first < batch ? n - first : batch, 1, s));
        MDC_CHECK(mdc_trainer_evaluate(t, x_dev + (size_t)n * 256, y_dev + (size_t)n * 3, nv, NULL, 0, nv, s));      /* validation_data  cnn.py:140 */
        double tl = 0, vl = 0;
        int64_t tf = 0, vf = 0, it = 0;
        MDC_CHECK(mdc_trainer_read(t, 1, &tl, &tf, &vl, &vf, &it, s));                        /* the epoch's one synchronisation */
        const double rec[2] = {tf ? tl / (double)tf : 0.0, vf ? vl / (double)vf : 0.0};
        fwrite(rec, sizeof(double), 2, out);
        printf("Epoch %d/%d - loss: %.6f - val_loss: %.6f (Adam step %lld)\n", ep + 1, epochs, rec[0], rec[1], (long long)it);
        if (rec[1] < best) {                                                                  /* ModelCheckpoint(save_best_only)  cnn.py:143 */
            best = rec[1];
            wait = 0;
            float* d = best_w;
            for (int l = 0; l < 2; ++l) {
                MDC_CHECK(mdc_trainer_get_tensor(t, MDC_TRAIN_WEIGHTS, l, d, nk[l], d + nk[l], nb[l], s));
                d += nk[l] + nb[l];
            }
        } else {
            ++wait;                                                                           /* EarlyStopping(patience=5)        cnn.py:144 */
        }
    }
    /* pad the record to `epochs` rows so that the file has a fixed shape, then the best weights */
    for (int ep = ran; ep < epochs; ++ep) { const double rec[2] = {-1.0, -1.0}; fwrite(rec, sizeof(double), 2, out); }
    fwrite(best_w, sizeof(float), nw, out);
    fclose(out);
    printf("trained %d epoch(s), best val_loss %.6f\n", ran, best);
    mdc_trainer_destroy(t);
    HIP_CHECK(hipFree(x_dev)); HIP_CHECK(hipFree(y_dev)); HIP_CHECK(hipFree(order_dev));
    HIP_CHECK(hipStreamDestroy(s));
    free(w); free(x); free(y); free(order); free(best_w);
    return 0;
}
