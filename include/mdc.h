/* mdc.h -- C ABI of libmdc.so: MI355X (gfx950) inference path for the VT-CNN2-family
 * modulation classifiers of peteroh23/ModulationDetectionCNN.
 *
 * The reference has no FFI/plugin interface; its boundary is the Keras object surface
 * (citations into /root/reference):
 *     model = models.Sequential(); model.add(...)      cnn.py:104-115, CNN.ipynb cell 6,
 *                                                      RML2016.10a_VTCNN2_example.ipynb:229-243
 *     model.load_weights(filepath)                     cnn.py:147, CNN.ipynb cell 8
 *     model.predict(X, batch_size=...)                 cnn.py:198, cnn.py:237, CNN.ipynb cell 12
 *     Model(inputs, outputs=model.layers[i].output)    CNN.ipynb cell 17 (layer taps)
 *     int(np.argmax(test_Y_hat[i,:]))                  cnn.py:209 (first maximum wins)
 * Each entry point below names the call it replaces.  INTEGRATION.md shows the ctypes
 * binding (modulationdetectioncnn_amd/_cabi.py is that binding).
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * errno-style code and sets a thread-local message (mdc_last_error); nothing aborts,
 * nothing throws across the boundary.  All device buffers belong to the caller; the
 * library owns only the packed weights.  mdc_forward is asynchronous on the caller's HIP
 * stream and performs no device synchronisation.  A finalized model is immutable and
 * mdc_forward on it is re-entrant (one model per device; any number of streams and host threads; with
 * mdc_set_profiling on, the per-launch event lists are kept under a mutex, so that stays true).  Every entry point
 * that touches a device selects the model's device for the duration of the call and restores the caller's.
 * The library is built with C++ exceptions enabled internally: an allocation failure inside it (weight copies,
 * packing buffers, event lists) is caught at the boundary and returned as MDC_ENOMEM.
 */
#ifndef MDC_H
#define MDC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libmdc.so is built with -fvisibility=hidden: the entry points below are its ONLY dynamic symbols
 * (tests/test_cabi.py compares `nm -D --defined-only` with this header, both ways). */
#if defined(__GNUC__) || defined(__clang__)
#define MDC_API __attribute__((visibility("default")))
#else
#define MDC_API
#endif

#define MDC_ABI_VERSION 5   /* 2: mdc_forward_iq_u8 takes hop + workspace; mdc_confusion_binned, mdc_iq_u8_windows
                               3: mdc_topology.reserved[0] is a validated option word (no bit defined: must be 0);
                                  mdc_iq_u8_windows wants 2-byte aligned input; the library reads no environment
                               4: MDC_KIND_VTCNN2 at MDC_FP8 keeps E4M3 features (its workspace per frame shrinks from
                                  22,144 to 11,584 bytes: ask mdc_workspace_bytes); MDC_OPT_FP8_BF16_FEATURES restores
                                  ABI 3's numerics and workspace
                               5: the training step (mdc_trainer_*, mdc_train_batch): additive */

/* error codes (negative errno values) */
#define MDC_OK        0
#define MDC_EINVAL   (-22)
#define MDC_ENOMEM   (-12)
#define MDC_ENODEV   (-19)
#define MDC_ENOTSUP  (-95)
#define MDC_EIO      (-5)     /* a HIP runtime call failed; see mdc_last_error() */
#define MDC_ESTATE   (-1)     /* call out of order (e.g. forward before finalize) */

/* Topology families.  Input is always (n, 2, 128) float32 I/Q. */
enum {
    MDC_KIND_DEPLOYED = 1,  /* CNN.ipynb cell 6: pad(0,1) Conv2D(F,(1,2)) relu Flatten Dense(C,relu) softmax;
                               F = filters in {3,10}, C = 3.  Layers: 0 conv, 1 dense.             */
    MDC_KIND_VTCNN2   = 2,  /* RML2016.10a_VTCNN2_example.ipynb:229-243: pad(0,2) Conv(256,1x3) relu
                               pad(0,2) Conv(80,2x3) relu Flatten Dense(256,relu) Dense(C) softmax.
                               Layers: 0 conv1, 1 conv2, 2 dense1, 3 dense2.  C <= 16.             */
    MDC_KIND_CNNPY    = 3   /* cnn.py:104-115 literal model (H=1,W=2,C=128 under channels_last):
                               pad(0,1) Conv2D(F,(1,2)) relu Flatten Dense(D,relu) Dense(C) softmax.
                               Layers: 0 conv, 1 dense1, 2 dense2.                                 */
};

/* Arithmetic type of the matrix products (accumulation is always f32).  MDC_BF16: MDC_KIND_VTCNN2 (both convs'
 * and dense1's operands) and MDC_KIND_DEPLOYED (the dense layer's operands; the conv stays f32; no layer taps).
 * MDC_KIND_VTCNN2 in the 16-bit modes keeps its activations and features multiplied by 2^-32 internally (an exact
 * power of two; the ReLU then rides in the bf16 conversion's clamp bit): conv1 / conv2 activations of 2^32 (4.3e9)
 * and beyond saturate, those below 2^-94 flush to zero -- I/Q samples of order 1e-2 sit in the middle of that range.
 * MDC_KIND_DEPLOYED does the same with its conv taps and bias at MDC_F32 and MDC_BF16 (the ReLU rides in the second
 * fma's / the conversion's clamp bit; dense weights carry the 2^+32): bit-identity to Keras' operation order holds for
 * conv activations in [2^-94, 2^32) -- at 2^32 and beyond they saturate at 2^32 instead of growing on, below 2^-94 the
 * scaled value is an f32 denormal (bits lost, then zero), and an Inf activation yields finite probabilities where Keras
 * gives NaN (tests/test_deployed_gpu.py pins both ends).  The MDC_TAP_CONV / MDC_TAP_FLAT taps of the deployed nets come
 * from a kernel that reads the UNSCALED table and rectifies with fmaxf: outside that range a tap and the probabilities of
 * the same frame can disagree.
 * MDC_F16: MDC_KIND_DEPLOYED only -- as MDC_BF16 there, with IEEE f16 operands and the conv itself in packed f16
 * (11 significant bits instead of 8, but conv outputs must stay below 65,504).
 * MDC_FP8: MDC_KIND_VTCNN2 -- conv2 on the block-scaled e4m3 MFMA, its features handed to dense1 as E4M3 bytes (conv1
 * and dense1's arithmetic as in MDC_BF16; see MDC_OPT_FP8_BF16_FEATURES) -- and
 * MDC_KIND_DEPLOYED -- as MDC_BF16 there with e4m3 operands (BASELINE configs[4] read literally); the activations are
 * scaled for the largest |sample| given with mdc_set_fp8_input_absmax -- beyond it they saturate. */
enum { MDC_F32 = 0, MDC_BF16 = 1, MDC_FP8 = 2, MDC_F16 = 3 };

/* Layer taps of CNN.ipynb cell 17.  tap_dev receives, per frame:
 *   MDC_TAP_CONV   model4: last conv+ReLU output in the reference's layout
 *                  (deployed: (2,129,F) channels_last; vtcnn2: (80,132) channels_first)
 *   MDC_TAP_FLAT   model3: Flatten output (same values, same order as CONV)
 *   MDC_TAP_DENSE  model2: output of the last Dense before softmax
 *                  (deployed: post-ReLU (C); vtcnn2/cnnpy: logits (C))
 *   MDC_TAP_HIDDEN vtcnn2/cnnpy only: Dense1+ReLU output                                   */
enum { MDC_TAP_NONE = 0, MDC_TAP_CONV = 1, MDC_TAP_FLAT = 2, MDC_TAP_DENSE = 3, MDC_TAP_HIDDEN = 4 };

typedef struct mdc_topology {
    int32_t kind;        /* MDC_KIND_*                                                       */
    int32_t filters;     /* deployed: F (3 or 10); cnnpy: F (10); vtcnn2: ignored (256/80)    */
    int32_t hidden;      /* cnnpy: D (10); vtcnn2: ignored (256); deployed: ignored           */
    int32_t classes;     /* C                                                                */
    int32_t reserved[4]; /* [0]: option bits MDC_OPT_* (0 = defaults); [1..3] must be 0      */
} mdc_topology;

/* Option bits (mdc_topology.reserved[0]; validated by mdc_create, fixed for the model's life).  Every kernel evaluates
 * the layers in Keras' operation order whatever the bits say (round 2's re-associated conv of the 10-filter net and the
 * bit that switched it off are gone: the ReLU now rides in the fma's clamp bit, which is faster AND exact).
 * MDC_OPT_FP8_BF16_FEATURES (MDC_KIND_VTCNN2, honoured at MDC_FP8 only, refused at any other dtype): keep the conv2
 * features between the conv kernel and dense1 in bf16, as the library did up to ABI 3.  Default since ABI 4: E4M3 bytes
 * with one power-of-two scale per tensor (half the feature traffic of both kernels; dense1 converts them back to bf16
 * exactly and keeps bf16 weights, so only the features' own rounding -- 3 significant bits + hidden one instead of
 * 7 + 1 -- changes: tests/test_label_agreement_gpu.py holds the same label floors for both). */
#define MDC_OPT_FP8_BF16_FEATURES 1
#define MDC_OPT_ALL 1

typedef struct mdc_model mdc_model;   /* opaque, owned by the library */

MDC_API int mdc_abi_version(void);

/* models.Sequential() + model.add(...): describe the net.  `device` is the HIP ordinal. */
MDC_API int mdc_create(const mdc_topology* topo, int device, mdc_model** out);

/* Number of weighted layers and the element counts load_weights must supply for each. */
MDC_API int mdc_num_layers(const mdc_model* m);
MDC_API int mdc_layer_sizes(const mdc_model* m, int layer, size_t* kernel_elems, size_t* bias_elems);

/* model.load_weights(): one call per weighted layer, host pointers in the Keras layout
 * (deployed/cnnpy conv: HWIO; vtcnn2 conv: OIHW; dense: (in, out) with `in` in the
 * reference's Flatten order).  The data is copied; the caller keeps ownership.            */
MDC_API int mdc_set_weights(mdc_model* m, int layer, const float* kernel_host, size_t kernel_elems,
                    const float* bias_host, size_t bias_elems);

/* Pack the weights into the kernels' register/LDS/MFMA layouts and upload them.
 * After this the model is immutable.  dtype: MDC_F32 | MDC_BF16 | MDC_FP8. */
MDC_API int mdc_finalize(mdc_model* m, int dtype);

/* MDC_FP8 only, before mdc_finalize: the largest |I/Q sample| the caller will feed (default 0.02, the scale of the
 * reference's bundled frames).  It fixes the power-of-two scale of the fp8 activations. */
MDC_API int mdc_set_fp8_input_absmax(mdc_model* m, float absmax);

/* MDC_KIND_VTCNN2 at MDC_FP8 with E4M3 features (the default), before mdc_finalize: the largest conv2 + ReLU output the
 * features must represent, e.g. twice the largest MDC_TAP_CONV value a bf16 / f32 forward of a sample batch produced
 * (calibration).  It fixes the power-of-two scale of the E4M3 bytes: values up to 2 x absmax keep their e4m3 precision,
 * larger ones saturate.  Not called: the library estimates it from the weights (12 sigma of the conv2 sum for independent
 * samples of rms fp8_input_absmax / 4).  Added in ABI 5. */
MDC_API int mdc_set_fp8_feature_absmax(mdc_model* m, float absmax);

/* Bytes of caller-owned device scratch ONE mdc_forward call of n frames needs (0 for deployed / cnnpy).
 * MDC_KIND_VTCNN2 keeps a call's conv2 features there: n rounded up to 256 frames x (10,560 features x 2 B in the
 * bf16 / fp8 modes, 4 B at f32, + 1 KiB of hidden layer) = 22,144 B (f32: 43,264 B) per frame -- 1.45 GB for a
 * 65,536-frame call, 23.2 GB for a 2^20-frame one.  The cost is PER CONCURRENT CALL: forwards enqueued on different
 * streams must not share a workspace.  A caller bounds it by splitting a batch into several calls (results do not
 * depend on the split; 65,536 frames per call already fill the chip 16 times over and cost 1.5 % against one
 * 2^20-frame call); modulationdetectioncnn_amd.VTCNN2 does that by default. */
MDC_API size_t mdc_workspace_bytes(const mdc_model* m, int64_t n);

/* model.predict(X) (+ np.argmax): x_dev (n,2,128) f32 contiguous on the model's device.
 * probs_dev (n,C) f32 or NULL; labels_dev (n) int32 or NULL (first-max tie-break);
 * tap_dev NULL unless tap != MDC_TAP_NONE.  Enqueued on hip_stream (NULL = null stream). */
MDC_API int mdc_forward(const mdc_model* m, const void* x_dev, int64_t n,
                float* probs_dev, int32_t* labels_dev,
                float* tap_dev, int tap,
                void* workspace_dev, size_t workspace_bytes,
                void* hip_stream);

/* ---- callers either side of the forward -------------------------------------------

 * Q6.12 integer forward of a DEPLOYED model: the arithmetic of the reference's FPGA datapath
 * (cnn_test_latest1.sv:642-675 bit selections, 18-bit wrap, 32-bit dense accumulators, sv:293-343, 171-176), for
 * validating ROM tables / test vectors written by the float2fix exporter (CNN.ipynb cells 23-24).  The integer
 * tables are made from the model's weights at mdc_finalize (float2fix: trunc(v * 4096)).
 * x_dev: (n,2,128) f32 frames (quantised on load, x_is_q612 = 0) or int32 Q6.12 words (x_is_q612 = 1).
 * dense_dev (n,C) int32 = post-ReLU class sums, value/4096 (or NULL); labels_dev (n) int32 first maximum (or NULL). */
MDC_API int mdc_forward_q612(const mdc_model* m, const void* x_dev, int x_is_q612, int64_t n,
                     int32_t* dense_dev, int32_t* labels_dev, void* hip_stream);

/* Confusion counts of cnn.py:205-216 / 242-255 on the device: counts_dev[t*classes + p] += 1 for every i with
 * truth_dev[i] == t and pred_dev[i] == p (int64, caller-zeroed, accumulates across calls).  Pairs with a label
 * outside [0,classes) are added to *bad_dev instead (may be NULL).  classes <= 32.  Runs on the current device. */
MDC_API int mdc_confusion(const int32_t* truth_dev, const int32_t* pred_dev, int64_t n, int classes,
                  int64_t* counts_dev, int64_t* bad_dev, void* hip_stream);

/* The per-SNR evaluation loop of cnn.py:228-259 in ONE launch: bin_dev[i] in [0,bins) says which SNR (or any other
 * grouping) frame i belongs to, and counts_dev[(b*classes + t)*classes + p] += 1 -- a (bins, classes, classes) int64
 * histogram, caller-zeroed; acc[b] = trace / sum of slice b (cnn.py:257-259).  Entries with a label or bin out of
 * range go to *bad_dev (may be NULL).  classes <= 32, bins <= 65536.  Runs on the current device. */
MDC_API int mdc_confusion_binned(const int32_t* truth_dev, const int32_t* pred_dev, const int32_t* bin_dev, int64_t n,
                         int classes, int bins, int64_t* counts_dev, int64_t* bad_dev, void* hip_stream);

/* score = model.evaluate(X_test, Y_test, ...)  (cnn.py:153; the reference compiles with loss='categorical_crossentropy' and
 * no metric, so the score is the mean loss): *loss_sum_dev += sum over i of -log(clip(p_i[t_i] / sum_c p_i[c], 1e-7, 1 - 1e-7))
 * with p_i = probs_dev[i*classes ..] (the softmax rows mdc_forward returned) and t_i = truth_dev[i] (the index of the
 * one-hot row's 1) -- Keras' categorical_crossentropy on probabilities.  f64, caller-zeroed, accumulates across calls; the
 * mean is loss_sum / n.  Labels outside [0,classes) are counted in *bad_dev (may be NULL) and add nothing.  classes <= 32.
 * Runs on the current device.  Added in ABI 4 (additive). */
MDC_API int mdc_crossentropy(const float* probs_dev, const int32_t* truth_dev, int64_t n, int classes,
                     double* loss_sum_dev, int64_t* bad_dev, void* hip_stream);

/* Raw SDR bytes -> frames: iq_dev holds n frames of 128 interleaved unsigned 8-bit (I,Q) pairs (256 B/frame, the
 * RTL-SDR format of the front-end in the reference's README.md:5); x_dev (n,2,128) f32 receives
 * ((byte - 127.5) * scale) with I in row 0 and Q in row 1.  Runs on the current device. */
MDC_API int mdc_iq_u8_to_frames(const uint8_t* iq_dev, int64_t n, float scale, float* x_dev, void* hip_stream);

/* Sliding windows over one contiguous capture (the live RTL-SDR stream of README.md:5): window i holds the 128 (I,Q)
 * pairs starting at pair i*hop, i.e. bytes [2*hop*i, 2*hop*i + 256) of iq_dev, which must hold 2*hop*(n-1) + 256 bytes.
 * hop = MDC_HOP_FRAME (128) is mdc_iq_u8_to_frames.  x_dev (n,2,128) f32.  Runs on the current device. */
#define MDC_HOP_FRAME 128
MDC_API int mdc_iq_u8_windows(const uint8_t* iq_dev, int64_t n, int64_t hop, float scale, float* x_dev, void* hip_stream);
/* (iq_dev 2-byte aligned -- whole (I,Q) pairs --, as mdc_forward_iq_u8 requires; otherwise MDC_EINVAL) */

/* Conversion and forward in ONE pass, for the deployed nets (any of their dtypes) and the VT-CNN2 family (f32, bf16,
 * fp8): the forward kernel itself reads the raw bytes (the deployed kernels' loads / LDS-DMA, the VT-CNN2 conv
 * kernels' frame staging) and converts in registers with the arithmetic of mdc_iq_u8_to_frames, so probs/labels are
 * bit-identical to mdc_iq_u8_windows followed by mdc_forward -- with 2*hop (<= 256) instead of 1,024 B of HBM input
 * per window and no frame buffer in between.  n windows, `hop` pairs apart as in mdc_iq_u8_windows (MDC_HOP_FRAME:
 * disjoint 256-byte frames); iq_dev 2-byte aligned (windows at odd hops are not better aligned than that anyway: the
 * kernels use gfx950's unaligned global loads), 2*hop*(n-1) + 256 bytes.  probs_dev (n,C) f32 and labels_dev (n)
 * int32 may each be NULL.  workspace: as mdc_forward (mdc_workspace_bytes(m, n); NULL/0 for the deployed nets).
 * MDC_KIND_CNNPY: MDC_ENOTSUP (use the two calls).  This is the SDR -> classifier hand-off of README.md:5. */
MDC_API int mdc_forward_iq_u8(const mdc_model* m, const uint8_t* iq_dev, int64_t n, int64_t hop, float scale,
                      float* probs_dev, int32_t* labels_dev,
                      void* workspace_dev, size_t workspace_bytes, void* hip_stream);

/* test_Y_hat = model.predict(X_test, batch_size=batch_size)  (cnn.py:198, 237) when X_test lies in HOST memory -- a
 * numpy array, or whatever buffer a cgo / JNI / N-API caller holds: the library's own driver in front of mdc_forward.
 * Frames are copied into a pinned ring by a few host threads, DMA'd, computed and the results DMA'd back in three
 * overlapping slots of chunk_frames frames (0 = 65,536) on the library's own streams, so the call runs at
 * max(PCIe, kernel) rather than their sum; x_host that is already pinned (hipHostMalloc / hipHostRegister) is DMA'd
 * from where it lies.  SYNCHRONOUS, like Keras' predict: returns when probs_host (n,C) f32 and labels_host (n) int32
 * (each may be NULL) are complete.  Results are bit-identical to mdc_forward on the same frames, whatever the chunk.
 * Staging buffers, streams and the workspace belong to the model (created on first use, freed by mdc_destroy); calls
 * on one model are serialised.  Added in ABI 2 (additive). */
MDC_API int mdc_predict_host(mdc_model* m, const float* x_host, int64_t n, float* probs_host, int32_t* labels_host,
                     int64_t chunk_frames);

/* The same for raw uint8 I/Q in host memory (an SDR capture buffer, README.md:5): n windows, `hop` pairs apart, from
 * iq_host (2*hop*(n-1) + 256 bytes) through mdc_forward_iq_u8 -- 2*hop bytes per window over PCIe instead of 1,024. */
MDC_API int mdc_predict_host_iq_u8(mdc_model* m, const uint8_t* iq_host, int64_t n, int64_t hop, float scale,
                           float* probs_host, int32_t* labels_host, int64_t chunk_frames);

/* Measurement support (bench.py roofline leg): when on, mdc_forward brackets each kernel
 * launch with HIP events on the launch stream; mdc_profile_read synchronises on them and
 * returns the summed device time and launch count of kernel slot `slot` since the last
 * mdc_profile_reset.  Off by default; never on in the timed region of the headline number.
 * Forwards may run concurrently with profiling on (the event lists are mutex-guarded); mdc_set_profiling /
 * mdc_profile_reset themselves must not race with forwards of the same model. */
MDC_API int mdc_set_profiling(mdc_model* m, int on);
MDC_API int mdc_profile_slots(const mdc_model* m);
MDC_API const char* mdc_profile_name(const mdc_model* m, int slot);
MDC_API int mdc_profile_read(mdc_model* m, int slot, double* total_ms, int64_t* launches);
MDC_API int mdc_profile_reset(mdc_model* m);

/* ---- training -----------------------------------------------------------------------
 * The reference trains two nets: CNN.ipynb cell 6's (MDC_KIND_DEPLOYED; the five bundled .h5 files are its results) and
 * cnn.py:104-112's (MDC_KIND_CNNPY).  Neither contains a Dropout layer, so the training forward is the inference
 * forward.  MDC_KIND_VTCNN2 is refused (MDC_ENOTSUP): its training exists only in the vendored DeepSig notebook.  f32.
 *
 *     model.compile(loss='categorical_crossentropy', optimizer='adam')     cnn.py:113     mdc_trainer_create (+ _set_adam)
 *     (the freshly initialised / loaded weights)                            cnn.py:108-111 mdc_trainer_set_tensor(MDC_TRAIN_WEIGHTS)
 *     model.fit(...): one mini-batch of one epoch                           cnn.py:135     mdc_train_batch
 *     ... validation_data=(X_test, Y_test): val_loss after each epoch       cnn.py:140     mdc_trainer_evaluate + mdc_trainer_read
 *     ModelCheckpoint(filepath, save_best_only=True) / load_weights         cnn.py:143-147 mdc_trainer_get_tensor (weights, Adam m / v,
 *                                                                                          iterations: all a Keras full-model .h5 holds)
 * The loss is Keras 2.4's categorical_crossentropy on the softmax OUTPUT (the model ends in Activation('softmax') +
 * Reshape, so the loss sees probabilities): q = p / sum(p), clipped to [1e-7, 1 - 1e-7], L = -sum_c y_c log q_c, mean over
 * the batch; the clip passes no gradient outside its interval, a ReLU none where its input is <= 0.  The optimizer is
 * TensorFlow 2.4's Adam: alpha = lr sqrt(1 - beta2^t) / (1 - beta1^t), m += (g - m)(1 - beta1), v += (g g - v)(1 - beta2),
 * w -= m alpha / (sqrt(v) + eps) with t = iterations + 1 (eps is NOT scaled by the bias correction).
 * Everything below runs on the trainer's device; x_dev (frames (., 2, 128) f32, 16-byte aligned) and y_dev (target rows
 * (., classes) f32: the one-hot rows of cnn.py:74-82, or any distribution) are the caller's buffers holding the WHOLE set;
 * a mini-batch is the frames order_dev[first .. first + count) of it (order_dev: int32 indices on the device, the
 * epoch's shuffle -- frames are never moved; NULL = the identity).  n_frames is how many frames the two buffers hold: a
 * position whose index lies outside [0, n_frames) is SKIPPED on the device -- it adds nothing to the loss or the gradient
 * (the batch mean still divides by count) -- and is counted; the next mdc_trainer_read then returns MDC_EINVAL with the count:
 * a bad shuffle is an error at the epoch's read, never a GPU fault.  mdc_train_batch and mdc_trainer_evaluate only enqueue
 * on hip_stream (two launches, no synchronisation, no allocation: capturable in a hipGraph; Adam's step count lives on the
 * device).  A step is reproducible bit for bit (fixed-order reductions, no float atomics).  One stream at a time per trainer. */
typedef struct mdc_trainer mdc_trainer;   /* opaque: f32 master weights in the Keras layouts, Adam state, scratch */

MDC_API int mdc_trainer_create(const mdc_topology* topo, int device, mdc_trainer** out);
MDC_API int mdc_trainer_num_layers(const mdc_trainer* t);
MDC_API int mdc_trainer_layer_sizes(const mdc_trainer* t, int layer, size_t* kernel_elems, size_t* bias_elems);

/* keras.optimizers.Adam(lr, beta_1, beta_2, epsilon); the defaults (1e-3, 0.9, 0.999, 1e-7 -- what 'adam' at cnn.py:113
 * means and what every bundled .h5's training_config records) hold until this is called. */
MDC_API int mdc_trainer_set_adam(mdc_trainer* t, float lr, float beta1, float beta2, float eps);

/* OPTIONAL Dropout(rate), off by default (rate 0 = the nets as the reference defines them: cnn.py:104-112 and CNN.ipynb cell 6
 * contain no Dropout layer; their `dr = 0.5` / `0.6` is left over from the DeepSig definition, which has `model.add(Dropout(dr))`
 * behind every conv and behind dense1 -- RML2016.10a_VTCNN2_example.ipynb:229-243).  With rate > 0 the training batches
 * (mdc_train_batch, either value of `apply`) multiply the conv + ReLU output -- and, for MDC_KIND_CNNPY, the Dense(D, relu)
 * output -- by mask / (1 - rate), Keras' Dropout; mdc_trainer_evaluate and every mdc_forward* never do (inference).  TensorFlow's
 * generator cannot be replayed, so the mask comes from a stated counter-based one (uint32 arithmetic):
 *     fmix32(h):  h ^= h >> 16;  h *= 0x85EBCA6B;  h ^= h >> 13;  h *= 0xC2B2AE35;  h ^= h >> 16
 *     k_step  = fmix32(seed + 0x9E3779B9 * (iterations + 1))          iterations: Adam's step count (on the device)
 *     k_frame = fmix32(k_step ^ (frame * 0x85EBCA6B + site))          frame: the frame's index in x_dev; site 0 conv, 1 dense
 *     keep element e (its index in the layer's Flatten order) iff fmix32(k_frame + e * 0xC2B2AE35) >= floor(rate * 2^32)
 * -- a new mask at every step (also under hipGraph replay), the same mask for a frame whatever batch it arrives in. */
MDC_API int mdc_trainer_set_dropout(mdc_trainer* t, float rate, uint32_t seed);

/* Per-layer tensors in the layouts of mdc_set_weights.  MDC_TRAIN_WEIGHTS must be set for every layer before the first
 * batch; Adam's moments start at zero and `iterations` at 0 (set them to resume from a full-model .h5, whose
 * /optimizer_weights group holds exactly these).  MDC_TRAIN_GRADIENT (get only): d(mean loss)/d(weights) of the last
 * mdc_train_batch.  Both calls synchronise hip_stream (the host buffers are complete / reusable on return). */
enum { MDC_TRAIN_WEIGHTS = 0, MDC_TRAIN_ADAM_M = 1, MDC_TRAIN_ADAM_V = 2, MDC_TRAIN_GRADIENT = 3 };
MDC_API int mdc_trainer_set_tensor(mdc_trainer* t, int which, int layer, const float* kernel_host, size_t kernel_elems,
                           const float* bias_host, size_t bias_elems, void* hip_stream);
MDC_API int mdc_trainer_get_tensor(mdc_trainer* t, int which, int layer, float* kernel_host, size_t kernel_elems,
                           float* bias_host, size_t bias_elems, void* hip_stream);
MDC_API int mdc_trainer_set_iterations(mdc_trainer* t, int64_t iterations, void* hip_stream);

/* model.train_on_batch / one step of model.fit (cnn.py:135): forward, loss, backward over the `count` frames and, if
 * `apply` != 0, one Adam update (apply = 0: the gradient is computed and kept for MDC_TRAIN_GRADIENT, nothing changes).
 * The batch's summed loss and frame count are added to the trainer's running training statistics. */
MDC_API int mdc_train_batch(mdc_trainer* t, const float* x_dev, const float* y_dev, int64_t n_frames, const int32_t* order_dev,
                    int64_t first, int64_t count, int apply, void* hip_stream);

/* The val_loss half of model.fit's epoch end / model.evaluate with the weights as they are now (cnn.py:140, 153): adds the
 * summed per-sample loss and the frame count to the trainer's evaluation statistics. */
MDC_API int mdc_trainer_evaluate(mdc_trainer* t, const float* x_dev, const float* y_dev, int64_t n_frames, const int32_t* order_dev,
                         int64_t first, int64_t count, void* hip_stream);

/* Synchronise hip_stream and read the statistics (each pointer may be NULL): sums of per-sample losses and frame counts
 * since the last reset, for training batches and for evaluation (mean = sum / frames: fit's `loss` and `val_loss`), and
 * Adam's step count.  reset != 0 zeroes the four statistics (not `iterations`). */
MDC_API int mdc_trainer_read(mdc_trainer* t, int reset, double* train_loss_sum, int64_t* train_frames, double* eval_loss_sum,
                     int64_t* eval_frames, int64_t* iterations, void* hip_stream);

MDC_API void mdc_trainer_destroy(mdc_trainer* t);

MDC_API const char* mdc_last_error(void);
MDC_API void mdc_destroy(mdc_model* m);

#ifdef __cplusplus
}
#endif
#endif /* MDC_H */
