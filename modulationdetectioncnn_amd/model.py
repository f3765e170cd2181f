"""Host-side mirror of the reference's Keras surface for the inference path.

    reference (cnn.py / CNN.ipynb)                     here
    -----------------------------------------------    -----------------------------------
    model = models.Sequential(); model.add(...)        m = VTCNN2(Topology.deployed(3))
    model.load_weights(filepath)      cnn.py:147       m.load_weights(path)   (.h5 | .txt | .npz)
    model.compile(loss=..., optimizer='adam') :113     m.compile(loss='categorical_crossentropy', optimizer='adam')
    history = model.fit(X, Y, ...)    cnn.py:135       history = m.fit(X, Y, batch_size=1024, epochs=100, validation_data=...,
                                                                       checkpoint=filepath, patience=5)   (training.py)
    model.predict(X, batch_size=1024) cnn.py:198       m.predict(X, batch_size=1024)
    int(np.argmax(Y_hat[i,:]))        cnn.py:209       m.predict_classes(X)
    Model(inputs, layers[i].output)   CNN.ipynb c.17   m.predict(X, tap='conv'|'flat'|'dense'), or the cell's own spelling:
                                                       Model(inputs=m.inputs, outputs=m.layers[4].output).predict(X)

All arithmetic runs in libmdc.so (hand-written gfx950 HIP) through the C ABI in
include/mdc.h; PyTorch-ROCm only supplies device memory and the HIP stream.  numpy in ->
numpy out (PCIe-bound plumbing); torch-ROCm tensor in -> torch tensor out on the same device,
enqueued on torch's current stream with no synchronisation.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _cabi
from .formats import q612
from .formats.h5mini import load_keras_h5
from .topology import Topology, synthetic_weights

_KIND = {"deployed": _cabi.KIND_DEPLOYED, "vtcnn2": _cabi.KIND_VTCNN2, "cnnpy": _cabi.KIND_CNNPY}
_DTYPE = {"f32": _cabi.F32, "fp32": _cabi.F32, "float32": _cabi.F32, "bf16": _cabi.BF16, "bfloat16": _cabi.BF16,
          "fp8": _cabi.FP8, "f16": _cabi.F16, "fp16": _cabi.F16, "float16": _cabi.F16}
# slots of the host-buffer driver are at least this long (frames), whatever batch_size a Keras-style call passes
HOST_MIN_CHUNK = 16384

_TAP = {None: _cabi.TAP_NONE, "conv": _cabi.TAP_CONV, "flat": _cabi.TAP_FLAT, "dense": _cabi.TAP_DENSE,
        "hidden": _cabi.TAP_HIDDEN}

Weights = List[Tuple[np.ndarray, np.ndarray]]


def _torch():
    import torch
    return torch


class VTCNN2:
    """A VT-CNN2-family classifier bound to one MI355X."""

    # scratch buffers kept alive at once (one per stream that forwards concurrently; least recently used goes first)
    MAX_WORKSPACES = 4

    def __init__(self, topology: Topology, device: Union[int, str, None] = None, dtype: str = "f32",
                 fp8_input_absmax: Optional[float] = None, fp8_bf16_features: bool = False,
                 fp8_feature_absmax: Optional[float] = None, _lib_variant: str = "product"):
        """dtype "f32" | "bf16" (vtcnn2, deployed) | "f16" (deployed) | "fp8" (vtcnn2, deployed).  fp8_input_absmax: the largest
        |I/Q sample| the fp8 mode must represent (default 0.02, the scale of the reference's frames); larger inputs saturate.
        fp8_bf16_features (vtcnn2 at "fp8" only): MDC_OPT_FP8_BF16_FEATURES -- keep the conv2 features in bf16 as before
        ABI 4 instead of E4M3 bytes (twice the feature traffic, the old numerics).
        fp8_feature_absmax (vtcnn2 at "fp8", E4M3 features): the largest conv2 feature to represent (mdc_set_fp8_feature_absmax);
        None = the library's estimate from the weights; calibrate_fp8_features() measures it on a sample.
        _lib_variant: tests only (the alternates build)."""
        self.topology = topology
        self.dtype = dtype
        self.fp8_input_absmax = fp8_input_absmax
        self.fp8_bf16_features = bool(fp8_bf16_features)
        self._fp8_feature_absmax = fp8_feature_absmax
        if self.fp8_bf16_features and not (topology.kind == "vtcnn2" and dtype == "fp8"):
            raise ValueError("fp8_bf16_features is an option of the vtcnn2 family's fp8 mode")
        self._lib_variant = _lib_variant
        if dtype not in _DTYPE:
            raise ValueError(f"dtype must be one of {sorted(_DTYPE)}")
        # the device is resolved ONCE (None = the device current at construction): the engine handle, the workspace
        # and the input check all use this stored ordinal, whatever torch.cuda.set_device() does later
        self._device_index: Optional[int] = self._resolve_device(device)
        self._weights: Optional[Weights] = None
        self._handle: Optional[C.c_void_p] = None
        self._ws = {}
        self._ws_captured = []  # buffers whose address a captured hipGraph holds (never evicted; see _workspace)
        self._ws_need = {}      # frames per launch -> mdc_workspace_bytes (a property of the finalized model)
        # frames per mdc_forward call when the caller gives no batch_size.  VT-CNN2: the workspace holds one call's features
        # (21.6 KB/frame in the 16-bit modes, 42 KB at f32): 65,536 frames per call = 1.45 GB (f32: 2.8 GB) per stream, and
        # every launch still fills the chip 16 times over.  A caller with HBM to spare passes batch_size=1 << 20 (23 GB of
        # workspace, +1.5 %: one launch of each kernel instead of sixteen) -- bench.py's headline does, and says so.
        self.default_chunk = 1 << 16 if topology.kind == "vtcnn2" else 1 << 22

    # what the packed engine was built from cannot change under it
    _FIXED = ("topology", "dtype", "fp8_bf16_features", "_lib_variant")

    def __setattr__(self, name, value):
        if name in VTCNN2._FIXED and name in self.__dict__:
            raise AttributeError(f"{name} is fixed at construction: build another VTCNN2")
        if name == "fp8_input_absmax" and self.__dict__.get("_handle") is not None:
            self._release()                 # re-packed with the new activation scale at the next use
        object.__setattr__(self, name, value)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_h5(cls, path: str, **kw) -> "VTCNN2":
        ck = load_keras_h5(path)
        m = cls(Topology.from_keras_config(ck.model_config), **kw)
        m._set_from_checkpoint(ck)
        return m

    @classmethod
    def from_npz(cls, path: str, **kw) -> "VTCNN2":
        """Decoded-weights fixture (tests/golden/weights/*.npz, made by tools/make_golden.py)."""
        z = np.load(path)
        F = int(z["conv_bias"].shape[0])
        m = cls(Topology.deployed(F, int(z["dense_bias"].shape[0])), **kw)
        m.set_weights([(z["conv_kernel"], z["conv_bias"]), (z["dense_kernel"], z["dense_bias"])])
        return m

    @classmethod
    def from_txt(cls, path: str, conv_from: Optional[str] = None, strict: bool = False, **kw) -> "VTCNN2":
        """Q6.12 text export.  F=10 dumps hold only the dense kernel: ``conv_from`` names the .h5
        (or .npz fixture) supplying conv kernel/bias and dense bias."""
        w = q612.load_weights_txt(path, strict=strict)
        m = cls(Topology.deployed(w.filters, 3), **kw)
        donor = None
        if conv_from is not None:
            donor = (cls.from_npz(conv_from) if conv_from.endswith(".npz") else cls.from_h5(conv_from))._weights
        parts = {"conv_kernel": w.conv_kernel, "conv_bias": w.conv_bias, "dense_kernel": w.dense_kernel, "dense_bias": w.dense_bias}
        if w.placeholder_dense:
            parts["dense_kernel"] = None      # 12.14.weights.txt: six identical placeholder tables
        if donor is not None:
            fill = {"conv_kernel": donor[0][0], "conv_bias": donor[0][1], "dense_kernel": donor[1][0], "dense_bias": donor[1][1]}
            for k, v in fill.items():
                if parts[k] is None:
                    parts[k] = v
        missing = [k for k, v in parts.items() if v is None]
        if missing:
            raise ValueError(f"{os.path.basename(path)} lacks {missing}; pass conv_from=<.h5 or .npz>")
        m.set_weights([(parts["conv_kernel"], parts["conv_bias"]), (parts["dense_kernel"], parts["dense_bias"])])
        return m

    @classmethod
    def synthetic(cls, topology: Union[Topology, str] = "vtcnn2", classes: int = 11, seed: int = 2016,
                  bias_scale: float = 0.0, **kw) -> "VTCNN2":
        if isinstance(topology, str):
            topology = {"vtcnn2": Topology.vtcnn2(classes), "deployed3": Topology.deployed(3, 3),
                        "deployed10": Topology.deployed(10, 3), "cnnpy": Topology.cnnpy(10, 10, classes)}[topology]
        m = cls(topology, **kw)
        m.set_weights(synthetic_weights(topology, seed, bias_scale))
        return m

    # ------------------------------------------------------------------ weights
    def _set_from_checkpoint(self, ck) -> None:
        tensors = [a for l in ck.layer_names for _, a in ck.weights[l]]
        if len(tensors) != 2 * len(self.topology.layer_shapes):
            raise ValueError(f"{ck.path}: {len(tensors)} tensors for {len(self.topology.layer_shapes)} weighted layers")
        self.set_weights([(tensors[2 * i], tensors[2 * i + 1]) for i in range(len(tensors) // 2)])

    def load_weights(self, filepath: str, conv_from: Optional[str] = None) -> None:
        """``model.load_weights(filepath)`` (cnn.py:147): topology must match the file's."""
        if filepath.endswith((".h5", ".hdf5")):
            ck = load_keras_h5(filepath)
            t = Topology.from_keras_config(ck.model_config)
            if t != self.topology:
                raise ValueError(f"{filepath} holds {t}, model is {self.topology}")
            self._set_from_checkpoint(ck)
        elif filepath.endswith(".npz"):
            self.set_weights(VTCNN2.from_npz(filepath)._weights)
        elif filepath.endswith(".txt"):
            self.set_weights(VTCNN2.from_txt(filepath, conv_from=conv_from)._weights)
        else:
            raise ValueError(f"unknown weight file type: {filepath}")

    def set_weights(self, weights: Sequence[Tuple[np.ndarray, np.ndarray]], theano_kernels: bool = False) -> None:
        """[(kernel, bias)] per weighted layer in the layouts of `Topology.layer_shapes`.  The kernels are applied as
        Keras-2/TensorFlow applies them: cross-correlation.  `theano_kernels=True` is for a checkpoint trained with
        Keras 1 on the Theano backend, as the vendored DeepSig notebook was (RML2016.10a_VTCNN2_example.ipynb:229-243
        under `K.set_image_dim_ordering('th')`): Theano's conv2d CONVOLVES, i.e. applies each filter flipped along both
        spatial axes, so the 4-D kernels are flipped once here and everything below stays a correlation."""
        shapes = self.topology.layer_shapes
        if len(weights) != len(shapes):
            raise ValueError(f"expected {len(shapes)} (kernel, bias) pairs, got {len(weights)}")
        out = []
        for i, ((k, b), (ks, bs)) in enumerate(zip(weights, shapes)):
            k = np.ascontiguousarray(k, dtype=np.float32)
            b = np.ascontiguousarray(b, dtype=np.float32)
            if theano_kernels and k.ndim == 4:
                sp = (2, 3) if self.topology.kind == "vtcnn2" else (0, 1)      # OIHW / HWIO: the two spatial axes
                k = np.ascontiguousarray(np.flip(k, axis=sp))
            if tuple(k.shape) != ks or tuple(b.shape) != bs:
                raise ValueError(f"layer {i} ({self.topology.layer_names[i]}): expected kernel {ks} bias {bs}, got {k.shape} {b.shape}")
            out.append((k, b))
        self._weights = out
        self._release()
        t = getattr(self, "_trainer", None)
        if t is not None:           # as in Keras: set_weights / load_weights replace the layers' variables, the optimizer keeps its
            t.set_weights(out)      # slots (Adam's m, v, iterations) -- cnn.py:147 reloads the best epoch into a compiled model

    def get_weights(self) -> Weights:
        if self._weights is None:
            raise RuntimeError("no weights loaded")
        return [(k.copy(), b.copy()) for k, b in self._weights]

    # ------------------------------------------------------------------ training (cnn.py:113, 122-147)
    def compile(self, loss: str = "categorical_crossentropy", optimizer: str = "adam", lr: float = 1e-3, beta1: float = 0.9,
                beta2: float = 0.999, eps: float = 1e-7, dropout: float = 0.0, dropout_seed: int = 0) -> None:
        """``model.compile(loss='categorical_crossentropy', optimizer='adam')`` (cnn.py:113): the one loss and the one
        optimizer the reference uses; the keyword arguments are keras.optimizers.Adam's (its defaults).  dropout: the optional
        Dropout(dr) of the DeepSig definition behind the conv activations (training batches only; 0 = the reference's nets)."""
        if loss != "categorical_crossentropy" or str(optimizer).lower() != "adam":
            raise ValueError("the reference compiles with loss='categorical_crossentropy', optimizer='adam'; nothing else is built")
        self._adam = dict(lr=float(lr), beta1=float(beta1), beta2=float(beta2), eps=float(eps), dropout=float(dropout),
                          dropout_seed=int(dropout_seed))
        self._drop_trainer()

    def _drop_trainer(self) -> None:
        t = getattr(self, "_trainer", None)
        if t is not None:
            t.close()
        self._trainer = None

    def trainer(self):
        """The training state behind fit(): created on first use from the current weights; it lives on (as a compiled Keras
        model's optimizer does) until the weights are replaced from outside."""
        from .training import Trainer
        if getattr(self, "_trainer", None) is None:
            if self._weights is None:
                raise RuntimeError("set or load weights (or VTCNN2.synthetic) before fit()")
            self._trainer = Trainer(self.topology, self._weights, device=self.device_index, _lib_variant=self._lib_variant,
                                    **getattr(self, "_adam", {}))
        return self._trainer

    def fit(self, X, Y, batch_size: int = 1024, epochs: int = 100, validation_data=None, patience: Optional[int] = 5,
            checkpoint: Optional[str] = None, shuffle: bool = True, seed: Optional[int] = None, verbose: int = 0, **kw):
        """``model.fit(X_train, Y_train, batch_size, epochs, validation_data=(X_test, Y_test), callbacks=[ModelCheckpoint(
        checkpoint, monitor='val_loss', save_best_only=True), EarlyStopping(monitor='val_loss', patience=patience)])``
        (cnn.py:135-146; ``callbacks=[...]`` with the two classes of ``modulationdetectioncnn_amd.callbacks`` is accepted
        verbatim) for the deployed and cnn.py nets, f32, on the MI355X (csrc/train.hip).  As in Keras the model is
        left with the LAST epoch's weights; the best ones are in `checkpoint` (a Keras full-model .h5: ``load_weights`` it,
        as cnn.py:147 does) and in the returned history's ``best_weights``."""
        t = self.trainer()
        hist = t.fit(X, Y, batch_size=batch_size, epochs=epochs, validation_data=validation_data, patience=patience,
                     checkpoint=checkpoint, shuffle=shuffle, seed=seed, verbose=verbose, **kw)
        self._weights = t.get_weights()
        self._release()                     # the inference engine re-packs the new weights on its next use
        return hist

    def save(self, filepath: str) -> None:
        """``model.save(filepath)``: Keras 2.4 full-model HDF5 (formats/h5mini.write_keras_h5); Adam's state is included when
        this model has been fitted."""
        from .formats.h5mini import write_keras_h5
        if self._weights is None:
            raise RuntimeError("no weights loaded")
        t = getattr(self, "_trainer", None)
        write_keras_h5(filepath, self.topology, self._weights, optimizer=t.optimizer_state() if t is not None else None,
                       adam={k: v for k, v in (getattr(self, "_adam", None) or {}).items() if not k.startswith("dropout")})

    # ------------------------------------------------------------------ engine
    def _lib(self):
        return _cabi.lib(self._lib_variant)

    def _check(self, rc: int) -> int:
        return _cabi.check(rc, self._lib_variant)

    @staticmethod
    def _resolve_device(device) -> Optional[int]:
        """int | "cuda:N" | torch.device | None.  None is resolved when a GPU is present (at construction, else at
        first use: models are also built on GPU-less hosts to read or convert weights)."""
        if isinstance(device, (int, np.integer)):
            return int(device)
        torch = _torch()
        if device is None:
            return torch.cuda.current_device() if torch.cuda.is_available() else None
        d = torch.device(device)
        if d.type != "cuda":
            raise ValueError(f"the MI355X path needs a cuda (ROCm) device, got {device!r}")
        return d.index if d.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else None)

    @property
    def device_index(self) -> int:
        if self._device_index is None:
            self._device_index = _torch().cuda.current_device()
        return self._device_index

    def _engine(self) -> C.c_void_p:
        if self._handle is not None:
            return self._handle
        if self._weights is None:
            raise RuntimeError("load_weights() before predict()")
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device visible: the MI355X path has no CPU fallback")
        L = self._lib()
        t = self.topology
        topo = _cabi.MdcTopology(_KIND[t.kind], t.filters, t.hidden, t.classes,
                                 (C.c_int32 * 4)(_cabi.MDC_OPT_FP8_BF16_FEATURES if self.fp8_bf16_features else 0, 0, 0, 0))
        h = C.c_void_p()
        self._check(L.mdc_create(C.byref(topo), self.device_index, C.byref(h)))
        try:
            for i, (k, b) in enumerate(self._weights):
                self._check(L.mdc_set_weights(h, i, k.ctypes.data_as(C.POINTER(C.c_float)), k.size,
                                              b.ctypes.data_as(C.POINTER(C.c_float)), b.size))
            if self.fp8_input_absmax is not None:
                self._check(L.mdc_set_fp8_input_absmax(h, float(self.fp8_input_absmax)))
            if self.fp8_feature_absmax is not None and t.kind == "vtcnn2" and self.dtype == "fp8" and not self.fp8_bf16_features:
                self._check(L.mdc_set_fp8_feature_absmax(h, float(self.fp8_feature_absmax)))
            self._check(L.mdc_finalize(h, _DTYPE[self.dtype]))
        except Exception:
            L.mdc_destroy(h)
            raise
        self._handle = h
        return h

    @property
    def fp8_feature_absmax(self) -> Optional[float]:
        """The caller's bound on the conv2 features for the E4M3 feature scale (None: the estimate from the weights).  Setting it
        re-packs the engine at its next use."""
        return self._fp8_feature_absmax

    @fp8_feature_absmax.setter
    def fp8_feature_absmax(self, value: Optional[float]) -> None:
        if value is not None and not float(value) > 0.0:
            raise ValueError("fp8_feature_absmax must be positive (or None)")
        self._fp8_feature_absmax = None if value is None else float(value)
        if getattr(self, "_handle", None) is not None:
            self._release()

    def calibrate_fp8_features(self, X, headroom: float = 2.0) -> float:
        """Calibration of the fp8 mode's E4M3 feature scale on a sample batch: the same weights in bf16 mode, the
        conv tap of X, `headroom` x its largest value -> mdc_set_fp8_feature_absmax at the next (re-)finalize.  Returns the
        value set.  X: a few hundred representative frames."""
        if not (self.topology.kind == "vtcnn2" and self.dtype == "fp8" and not self.fp8_bf16_features):
            raise ValueError("calibration applies to the vtcnn2 family's fp8 mode with E4M3 features")
        probe = VTCNN2(self.topology, device=self.device_index, dtype="bf16", _lib_variant=self._lib_variant)
        probe.set_weights(self._weights)
        top = float(np.asarray(probe.predict(np.asarray(X, np.float32), tap="conv")).max())
        probe._release()
        if not top > 0.0:
            raise ValueError("the sample produced no positive conv2 feature")
        self.fp8_feature_absmax = headroom * top      # (the setter releases the packed engine)
        return self.fp8_feature_absmax

    def _release(self) -> None:
        if self._handle is not None:
            self._lib().mdc_destroy(self._handle)
            self._handle = None
        self._ws = {}
        self._ws_captured = []
        self._ws_need = {}

    def __del__(self):
        try:
            self._release()
            self._drop_trainer()
        except Exception:
            pass

    def _workspace(self, n: int, stream_key: Optional[int] = None):
        need = self._ws_need.get(n)
        if need is None:
            need = self._ws_need[n] = int(self._lib().mdc_workspace_bytes(self._engine(), n))
        if need == 0:
            return None, 0
        torch = _torch()
        # one scratch buffer PER STREAM: forwards enqueued on different streams run concurrently and must not share
        # it (the C ABI leaves the workspace to the caller for exactly this reason); allocated under the stream
        # that uses it, so torch's caching allocator orders any reuse after the launches already queued there
        key = stream_key if stream_key is not None else torch.cuda.current_stream(torch.device("cuda", self.device_index)).cuda_stream
        ws = self._ws.pop(key, None)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=f"cuda:{self.device_index}")
        self._ws[key] = ws                      # (dict order = recency: re-inserted at the end)
        if torch.cuda.is_current_stream_capturing() and not any(w is ws for w in self._ws_captured):
            # a hipGraph being captured bakes this buffer's ADDRESS into its kernel nodes: the buffer must outlive every
            # replay, so neither the LRU below nor a later, larger chunk may hand it back to the allocator.
            # It stays referenced here until release_captured_workspaces() / the model goes away.
            self._ws_captured.append(ws)
        while len(self._ws) > self.MAX_WORKSPACES:
            # streams that are gone (worker threads' streams, a finished MultiStreamPredictor) do not pin HBM for ever:
            # the least recently used buffer goes back to torch's caching allocator, which keeps a block freed while
            # its stream still has work queued away from other streams until that work is done
            self._ws.pop(next(iter(self._ws)))
        return ws, need

    def reserve_workspace(self, frames_per_call: int) -> int:
        """Allocate (now, on torch's current stream) the scratch buffer a forward of `frames_per_call` frames per
        mdc_forward call needs, so that the first timed / captured forward allocates nothing; returns its bytes."""
        _ws, need = self._workspace(int(frames_per_call))
        return need

    def release_captured_workspaces(self) -> None:
        """Drop the references that keep scratch buffers alive for captured hipGraphs.  Call only after every graph
        captured from this model's forwards has been destroyed: a replay after this writes through a stale address."""
        self._ws_captured = []

    def build(self, input_shape=None) -> None:
        """``model.build()`` (cnn.py:114): the topology fixes every shape already."""

    def summary(self) -> None:
        """``model.summary()`` (cnn.py:115) prints the layer table."""
        print(self.topology.summary())

    # ------------------------------------------------------------------ model.layers / model.inputs (CNN.ipynb cells 15, 17)
    @property
    def layers(self) -> List["Layer"]:
        """The Sequential's layers in Keras' order (`print(model.layers[2])`, `model.layers[4].output`: CNN.ipynb cells 15, 17)."""
        rows = self.topology.keras_layers()
        names = self.topology.keras_layer_names()
        return [Layer(self, i, cls, name, shape, params) for i, ((cls, shape, params), (_role, name)) in enumerate(zip(rows, names))]

    @property
    def inputs(self) -> "ModelInputs":
        return ModelInputs(self)

    # ------------------------------------------------------------------ inference
    def tap_shape(self, tap: str) -> Tuple[int, ...]:
        t = self.topology
        if t.kind == "deployed":
            return {"conv": (2, 129, t.filters), "flat": (258 * t.filters,), "dense": (t.classes,)}[tap]
        if t.kind == "vtcnn2":
            return {"conv": (80, 132), "flat": (10560,), "dense": (t.classes,), "hidden": (256,)}[tap]
        return {"conv": (1, 3, t.filters), "flat": (3 * t.filters,), "dense": (t.classes,), "hidden": (t.hidden,)}[tap]

    def forward_device(self, x, probs=None, labels=None, tap: Optional[str] = None, tap_out=None,
                       batch_size: Optional[int] = None):
        """Enqueue the forward on torch's current stream.  x: contiguous float32 CUDA tensor (n,2,128).
        Pre-allocated outputs may be passed; returns (probs, labels, tap_out)."""
        torch = _torch()
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise TypeError("forward_device needs a torch tensor on the ROCm device")
        if x.dtype != torch.float32 or tuple(x.shape[1:]) != (2, 128) or not x.is_contiguous():
            raise ValueError(f"input must be contiguous float32 (n,2,128); got {x.dtype} {tuple(x.shape)}")
        if x.device.index != self.device_index:
            raise ValueError(f"input on cuda:{x.device.index}, model on cuda:{self.device_index}")
        if tap not in _TAP:
            raise ValueError(f"tap must be one of {[k for k in _TAP if k]}")
        n = x.shape[0]
        Cn = self.topology.classes
        h = self._engine()
        L = self._lib()
        dev = x.device
        if probs is None:
            probs = torch.empty((n, Cn), dtype=torch.float32, device=dev)
        if labels is None:
            labels = torch.empty((n,), dtype=torch.int32, device=dev)
        if tap is not None and tap_out is None:
            tap_out = torch.empty((n,) + self.tap_shape(tap), dtype=torch.float32, device=dev)
        if n == 0:
            return probs, labels, tap_out
        chunk = int(batch_size) if batch_size else self.default_chunk
        chunk = max(1, min(chunk, n))
        stream = torch.cuda.current_stream(dev).cuda_stream      # (2 us: looked up once per call)
        ws, ws_bytes = self._workspace(chunk, stream)
        tap_row = int(np.prod(self.tap_shape(tap))) if tap is not None else 0
        for s in range(0, n, chunk):
            m = min(chunk, n - s)
            self._check(L.mdc_forward(
                h, x.data_ptr() + s * 1024, m,
                probs.data_ptr() + s * Cn * 4, labels.data_ptr() + s * 4,
                (tap_out.data_ptr() + s * tap_row * 4) if tap is not None else None, _TAP[tap],
                ws.data_ptr() if ws is not None else None, ws_bytes, stream))
        return probs, labels, tap_out

    def _host_chunk(self, batch_size) -> int:
        """Keras' batch_size is a speed knob, not a result knob (results do not depend on it): the host-buffer driver
        keeps its slots at least HOST_MIN_CHUNK frames long, whatever the caller's cnn.py:198 passes (1024 there)."""
        return max(int(batch_size), HOST_MIN_CHUNK) if batch_size else 0

    def predict_host(self, X: np.ndarray, batch_size: Optional[int] = None, want_probs: bool = True,
                     want_labels: bool = True, out=None) -> Tuple[Optional[np.ndarray], Optional[np.ndarray]]:
        """numpy frames in, numpy (probs, labels) out through the library's own host-buffer driver (mdc_predict_host:
        pinned ring, copy / compute / result streams overlapped).  Bit-identical to the device path.  out = (probs,
        labels): write into the caller's C-contiguous float32 (n,C) / int32 (n,) arrays (slices of a larger batch's)."""
        a = np.ascontiguousarray(np.asarray(X), dtype=np.float32)
        if a.ndim != 3 or a.shape[1:] != (2, 128):
            raise ValueError(f"expected input of shape (n,2,128); got {a.shape}")
        n, Cn = a.shape[0], self.topology.classes
        if out is not None:
            probs, labels = out
            for arr, shape, dt in ((probs, (n, Cn), np.float32), (labels, (n,), np.int32)):
                if arr is not None and not (isinstance(arr, np.ndarray) and arr.shape == shape and arr.dtype == dt and arr.flags.c_contiguous
                                            and arr.flags.writeable):
                    raise ValueError(f"out arrays must be writeable C-contiguous {np.dtype(dt).name} of shape {shape}")
        else:
            probs = np.empty((n, Cn), np.float32) if want_probs else None
            labels = np.empty((n,), np.int32) if want_labels else None
        self._check(self._lib().mdc_predict_host(self._engine(), a.ctypes.data, n,
                                                 probs.ctypes.data if probs is not None else None,
                                                 labels.ctypes.data if labels is not None else None, self._host_chunk(batch_size)))
        return probs, labels

    def _run(self, X, batch_size, tap):
        torch = _torch()
        as_numpy = not isinstance(X, torch.Tensor)
        if as_numpy:
            a = np.ascontiguousarray(np.asarray(X), dtype=np.float32)
            if a.ndim != 3 or a.shape[1:] != (2, 128):
                raise ValueError(f"expected input of shape (n,2,128); got {a.shape}")
            x = torch.from_numpy(a).to(f"cuda:{self.device_index}")
        else:
            x = X
            if not x.is_cuda:
                x = x.to(f"cuda:{self.device_index}")
            x = x.to(torch.float32).contiguous()
        probs, labels, tap_out = self.forward_device(x, tap=tap, batch_size=batch_size)
        return as_numpy, probs, labels, tap_out

    def predict(self, X, batch_size: Optional[int] = None, tap: Optional[str] = None, verbose: int = 0):
        """``model.predict(X, batch_size)``: (n,C) float32 softmax rows (``verbose`` is accepted and says nothing); with ``tap`` the named
        intermediate layer of CNN.ipynb cell 17 instead.  Results do not depend on batch_size.  numpy in -> numpy out
        (through the streaming host-buffer driver); torch-ROCm tensor in -> tensor out on torch's current stream."""
        if tap is None and not isinstance(X, _torch().Tensor):
            return self.predict_host(X, batch_size, want_labels=False)[0]
        as_numpy, probs, _labels, tap_out = self._run(X, batch_size, tap)
        out = tap_out if tap is not None else probs
        return out.cpu().numpy() if as_numpy else out

    def predict_classes(self, X, batch_size: Optional[int] = None):
        """Row-wise ``np.argmax`` of predict(X) (cnn.py:209: first maximum wins), int32 (n,)."""
        if not isinstance(X, _torch().Tensor):
            return self.predict_host(X, batch_size, want_probs=False)[1]
        as_numpy, _probs, labels, _ = self._run(X, batch_size, None)
        return labels.cpu().numpy() if as_numpy else labels

    # ------------------------------------------------------------------ evaluation (cnn.py:198-216)
    def confusion_counts_device(self, X, labels_true, batch_size: Optional[int] = None):
        """Un-normalised C x C counts ``conf[true, predicted]`` (cnn.py:205-216) as an int64 CUDA tensor, computed
        on the device (mdc_confusion); labels outside [0,C) raise."""
        torch = _torch()
        pred = self.predict_classes(X if isinstance(X, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(np.asarray(X), dtype=np.float32)).to(f"cuda:{self.device_index}"), batch_size)
        dev = pred.device
        truth = torch.as_tensor(np.asarray(labels_true) if not isinstance(labels_true, torch.Tensor) else labels_true)
        truth = truth.to(device=dev, dtype=torch.int32).contiguous()
        if truth.shape != pred.shape:
            raise ValueError(f"labels_true has shape {tuple(truth.shape)}, predictions {tuple(pred.shape)}")
        Cn = self.topology.classes
        counts = torch.zeros((Cn, Cn), dtype=torch.int64, device=dev)
        bad = torch.zeros((1,), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            self._check(self._lib().mdc_confusion(truth.data_ptr(), pred.data_ptr(), pred.numel(), Cn, counts.data_ptr(),
                                                  bad.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        nbad = int(bad.item())                  # the one synchronising read
        if nbad:
            raise ValueError(f"{nbad} labels lie outside [0, {Cn})")
        return counts

    def evaluate(self, X, Y, batch_size: Optional[int] = None, verbose: int = 0) -> float:
        """``score = model.evaluate(X_test, Y_test, verbose=0, batch_size=...)`` (cnn.py:153): the reference compiles its model with
        loss='categorical_crossentropy' and no metric, so the score is the MEAN LOSS -- Keras' categorical cross-entropy
        on the softmax rows (row scaled to sum 1, clipped to [1e-7, 1 - 1e-7]).  Y: one-hot rows (n, C) as cnn.py:80-82
        builds them, or class indices (n,).  One forward, one reduction launch (mdc_crossentropy), one scalar read-back."""
        torch = _torch()
        Xt = X if isinstance(X, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(X), dtype=np.float32)).to(f"cuda:{self.device_index}")
        probs = self.predict(Xt, batch_size)
        dev = probs.device
        y = Y if isinstance(Y, torch.Tensor) else torch.as_tensor(np.asarray(Y))
        if y.ndim == 2:
            if tuple(y.shape) != tuple(probs.shape):
                raise ValueError(f"one-hot targets have shape {tuple(y.shape)}, predictions {tuple(probs.shape)}")
            y = y.argmax(dim=1)
        truth = y.to(device=dev, dtype=torch.int32).contiguous()
        if truth.shape != (probs.shape[0],):
            raise ValueError(f"targets have shape {tuple(truth.shape)}, predictions {tuple(probs.shape)}")
        n, Cn = probs.shape
        if n == 0:
            return float("nan")
        loss = torch.zeros((1,), dtype=torch.float64, device=dev)
        bad = torch.zeros((1,), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            self._check(self._lib().mdc_crossentropy(probs.data_ptr(), truth.data_ptr(), n, Cn, loss.data_ptr(), bad.data_ptr(),
                                                     torch.cuda.current_stream(dev).cuda_stream))
        both = torch.cat([loss, bad.to(torch.float64)]).cpu().numpy()      # the one synchronising read
        if int(both[1]):
            raise ValueError(f"{int(both[1])} labels lie outside [0, {Cn})")
        return float(both[0] / n)

    def confusion(self, X, labels_true, batch_size: Optional[int] = None, normalize: bool = True) -> np.ndarray:
        """``conf[j,k] += 1`` over (true j, predicted k) then row-normalise, as cnn.py:199-216 does with the
        output of ``model.predict``; rows without samples stay 0."""
        conf = self.confusion_counts_device(X, labels_true, batch_size).cpu().numpy().astype(np.float64)
        if not normalize:
            return conf
        s = conf.sum(axis=1, keepdims=True)
        return np.divide(conf, s, out=np.zeros_like(conf), where=s > 0)

    def accuracy(self, X, labels_true, batch_size: Optional[int] = None) -> float:
        """cnn.py:257-259: cor / (cor + ncor) from the un-normalised confusion counts."""
        conf = self.confusion(X, labels_true, batch_size, normalize=False)
        return float(np.trace(conf) / max(conf.sum(), 1.0))

    def accuracy_by_snr(self, X, labels_true, snrs, batch_size: Optional[int] = None) -> Tuple[Dict, Dict]:
        """The per-SNR loop of cnn.py:228-259: one forward over the whole batch, then ONE launch
        (mdc_confusion_binned: an S x C x C histogram keyed by each frame's SNR bin) and one device->host copy for all
        SNR values; acc[snr] = cor / (cor + ncor) (cnn.py:257-259).  Returns (acc, conf): dicts keyed by SNR value;
        conf[snr] is the un-normalised C x C count matrix (numpy)."""
        torch = _torch()
        x = X if isinstance(X, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(np.asarray(X), dtype=np.float32)).to(f"cuda:{self.device_index}")
        pred = self.predict_classes(x, batch_size)
        dev = pred.device
        truth = torch.as_tensor(np.asarray(labels_true) if not isinstance(labels_true, torch.Tensor) else labels_true)
        truth = truth.to(device=dev, dtype=torch.int32).contiguous()
        snr_t = torch.as_tensor(np.asarray(snrs) if not isinstance(snrs, torch.Tensor) else snrs).to(dev)
        if truth.shape != pred.shape or snr_t.shape != pred.shape:
            raise ValueError("labels_true and snrs need one entry per frame")
        Cn = self.topology.classes
        if pred.numel() == 0:
            return {}, {}
        values, bins = torch.unique(snr_t, sorted=True, return_inverse=True)      # SNR value -> bin index (plumbing)
        bins = bins.to(torch.int32).contiguous()
        S = int(values.numel())
        counts = torch.zeros((S, Cn, Cn), dtype=torch.int64, device=dev)
        bad = torch.zeros((1,), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            self._check(self._lib().mdc_confusion_binned(truth.data_ptr(), pred.data_ptr(), bins.data_ptr(), pred.numel(), Cn, S,
                                                         counts.data_ptr(), bad.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        host = torch.cat([counts.view(-1), bad]).cpu().numpy()      # the one synchronising copy
        if int(host[-1]):
            raise ValueError(f"{int(host[-1])} labels lie outside [0, {Cn})")
        c_all = host[:-1].reshape(S, Cn, Cn)
        acc, conf = {}, {}
        for i, val in enumerate(values.tolist()):
            conf[val] = c_all[i]
            acc[val] = float(np.trace(c_all[i])) / float(max(c_all[i].sum(), 1))
        return acc, conf

    @staticmethod
    def save_results(path: str, acc: Dict, tag: str = "CNN2", dr: float = 0.5) -> None:
        """The results file of cnn.py:262-264: `cPickle.dump(("CNN2", 0.5, acc), fd)` -- the tuple (model tag, dropout
        rate, {snr: accuracy}) that the reference's plotting cells read back.  Written with pickle protocol 2 (what
        Python 2's cPickle reads); keys and values are plain Python numbers."""
        import pickle
        clean = {(int(k) if float(k).is_integer() else float(k)): float(v) for k, v in acc.items()}
        with open(path, "wb") as fd:
            pickle.dump((str(tag), float(dr), clean), fd, protocol=2)

    @staticmethod
    def load_results(path: str) -> Tuple[str, float, Dict]:
        """Read back a results file: the (tag, dropout, {snr: accuracy}) tuple of save_results holds only str / float /
        int / dict / tuple, which unpickle without a single global -- so the loader REFUSES every global (class or
        function reference): a file that names one (any third party's pickle, a hostile one) raises instead of running
        code from it."""
        import pickle

        class _NoGlobals(pickle.Unpickler):
            def find_class(self, module, name):
                raise pickle.UnpicklingError(f"results files hold plain numbers and strings only; refusing {module}.{name}")

        with open(path, "rb") as fd:
            obj = _NoGlobals(fd).load()
        if not (isinstance(obj, tuple) and len(obj) == 3 and isinstance(obj[0], str) and isinstance(obj[1], (int, float))
                and isinstance(obj[2], dict)
                and all(isinstance(k, (int, float)) and isinstance(v, (int, float)) for k, v in obj[2].items())):
            raise ValueError(f"{path} is not a (tag, dropout, {{snr: accuracy}}) results file")
        tag, dr, acc = obj
        return tag, float(dr), acc

    # ------------------------------------------------------------------ FPGA arithmetic
    def predict_q612(self, X, as_float: bool = True):
        """The deployed net evaluated in the FPGA's Q6.12 integer arithmetic (mdc_forward_q612; rules of
        cnn_test_latest1.sv:642-675, 293-343).  X: float frames (n,2,128), quantised like the reference's table
        writer (`float2fix`: trunc(v*4096)), or an integer array/tensor of Q6.12 words.  Returns (dense, labels):
        the post-ReLU class sums as float (int/4096) or int32, and the first-max labels (int32)."""
        torch = _torch()
        if self.topology.kind != "deployed":
            raise ValueError("the Q6.12 datapath exists for the deployed nets only")
        as_numpy = not isinstance(X, torch.Tensor)
        t = torch.as_tensor(np.asarray(X)) if as_numpy else X
        if tuple(t.shape[1:]) != (2, 128):
            raise ValueError(f"expected input of shape (n,2,128); got {tuple(t.shape)}")
        is_q = not t.dtype.is_floating_point
        t = t.to(device=f"cuda:{self.device_index}", dtype=torch.int32 if is_q else torch.float32).contiguous()
        n, Cn = t.shape[0], self.topology.classes
        dense = torch.empty((n, Cn), dtype=torch.int32, device=t.device)
        labels = torch.empty((n,), dtype=torch.int32, device=t.device)
        self._check(self._lib().mdc_forward_q612(self._engine(), t.data_ptr() if n else None, int(is_q), n, dense.data_ptr(),
                                                 labels.data_ptr(), torch.cuda.current_stream(t.device).cuda_stream))
        out = dense.to(torch.float32) / 4096.0 if as_float else dense
        return (out.cpu().numpy(), labels.cpu().numpy()) if as_numpy else (out, labels)

    # ------------------------------------------------------------------ raw SDR bytes
    def predict_iq_u8(self, iq, scale: Optional[float] = None, batch_size: int = 0, hop: int = 128):
        """`predict` on raw RTL-SDR samples: iq holds unsigned bytes (I0,Q0,I1,Q1,...), each sample becomes
        (byte - 127.5) * scale (default 1/127.5).  Window i is the 128 (I,Q) pairs starting at pair i*hop of the
        capture: hop = 128 cuts it into disjoint 256-byte frames (then the byte count must be a multiple of 256),
        a smaller hop slides the classifier over a live stream (n = (pairs - 128) // hop + 1 windows).  The deployed
        nets and the VT-CNN2 family read the bytes in the forward kernels themselves (mdc_forward_iq_u8: 2*hop B of
        HBM input per window, no frame buffer); cnn.py's literal model converts on the device first
        (frames_from_iq_u8) -- the results are bit-identical either way.  Returns (probs, labels) as device tensors
        for a device tensor input, numpy arrays otherwise."""
        torch = _torch()
        from .frontend import DEFAULT_SCALE, frames_from_iq_u8, window_count
        scale = DEFAULT_SCALE if scale is None else float(scale)
        as_numpy = not isinstance(iq, torch.Tensor)
        if as_numpy and self.topology.kind != "cnnpy":      # host bytes: the library's streaming driver (mdc_predict_host_iq_u8)
            b = np.ascontiguousarray(np.asarray(iq, dtype=np.uint8)).reshape(-1)
            n, Cn = window_count(b.size, hop), self.topology.classes
            probs, labels = np.empty((n, Cn), np.float32), np.empty((n,), np.int32)
            self._check(self._lib().mdc_predict_host_iq_u8(self._engine(), b.ctypes.data, n, int(hop), scale, probs.ctypes.data,
                                                           labels.ctypes.data, self._host_chunk(batch_size)))
            return probs, labels
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(iq, dtype=np.uint8))) if as_numpy else iq
        if t.dtype != torch.uint8:
            raise TypeError(f"iq must be uint8, got {t.dtype}")
        t = t.to(f"cuda:{self.device_index}").contiguous().view(-1)
        n, Cn = window_count(t.numel(), hop), self.topology.classes
        if t.data_ptr() % 2:
            t = t.clone()       # a view into a larger buffer starting at an odd byte: the ABI wants whole (I,Q) pairs
        probs = torch.empty((n, Cn), dtype=torch.float32, device=t.device)
        labels = torch.empty((n,), dtype=torch.int32, device=t.device)
        if n == 0:
            return (probs.cpu().numpy(), labels.cpu().numpy()) if as_numpy else (probs, labels)
        if self.topology.kind == "cnnpy":
            x = frames_from_iq_u8(t, scale, hop=hop)
            self.forward_device(x, probs, labels, batch_size=batch_size or None)
        else:
            L, h = self._lib(), self._engine()
            chunk = max(1, min(int(batch_size) if batch_size else self.default_chunk, n))
            with torch.cuda.device(t.device):
                stream = torch.cuda.current_stream(t.device).cuda_stream
                ws, ws_bytes = self._workspace(chunk, stream)
                for s0 in range(0, n, chunk):
                    m = min(chunk, n - s0)
                    self._check(L.mdc_forward_iq_u8(h, t.data_ptr() + 2 * hop * s0, m, hop, scale,
                                                    probs.data_ptr() + s0 * Cn * 4, labels.data_ptr() + s0 * 4,
                                                    ws.data_ptr() if ws is not None else None, ws_bytes, stream))
        return (probs.cpu().numpy(), labels.cpu().numpy()) if as_numpy else (probs, labels)

    # ------------------------------------------------------------------ measurement hooks
    def set_profiling(self, on: bool) -> None:
        self._check(self._lib().mdc_set_profiling(self._engine(), int(on)))
        if on:
            self._check(self._lib().mdc_profile_reset(self._engine()))

    def read_profile(self) -> Dict[str, Tuple[float, int]]:
        L, h = self._lib(), self._engine()
        out = {}
        for i in range(L.mdc_profile_slots(h)):
            ms, cnt = C.c_double(), C.c_int64()
            self._check(L.mdc_profile_read(h, i, C.byref(ms), C.byref(cnt)))
            out[L.mdc_profile_name(h, i).decode()] = (ms.value, cnt.value)
        return out


# ----------------------------------------------------------------------------------------------------------------------
# CNN.ipynb cell 17 spelled as the notebook spells it:
#     model2 = Model(inputs = model.inputs, outputs = model.layers[4].output)   # dense + ReLU, no softmax
#     model2.compile(loss='categorical_crossentropy', optimizer='adam'); model2.predict(np.array([X_test[i],]))
# A sub-model is a VIEW of its parent: the same device weights, one forward with the layer's tap (MDC_TAP_*), the result in
# the shape Keras gives that layer's output.  Layers whose output the kernels never materialise (the input reshapes and
# paddings, VT-CNN2's first convolution -- conv1 feeds conv2 from registers) have no tap and say so.
# ----------------------------------------------------------------------------------------------------------------------
class ModelInputs:
    def __init__(self, model: VTCNN2):
        self.model = model

    def __repr__(self):
        return "[<input (None, 2, 128) float32>]"


class LayerOutput:
    def __init__(self, layer: "Layer"):
        self.layer = layer

    @property
    def shape(self):
        return self.layer.output_shape

    def __repr__(self):
        return f"<output of {self.layer!r} {self.layer.output_shape}>"


class Layer:
    """One entry of `model.layers`: what the reference prints, reads weights from and taps."""

    def __init__(self, model: VTCNN2, index: int, class_name: str, name: str, shape: Tuple[int, ...], params: int):
        self.model, self.index, self.class_name, self.name = model, index, class_name, name
        self.output_shape = (None,) + tuple(shape)
        self._params = params

    def __repr__(self):
        return f"<{self.class_name} {self.name}>"

    def count_params(self) -> int:
        return self._params

    def get_weights(self) -> List[np.ndarray]:
        """[kernel, bias] of a Conv2D / Dense in the layout Keras holds them, [] for the others (`layer.get_weights()`)."""
        if self._params == 0:
            return []
        rows = self.model.topology.keras_layers()
        k = sum(1 for _c, _s, p in rows[:self.index] if p)
        kernel, bias = self.model.get_weights()[k]
        return [kernel, bias]

    @property
    def output(self) -> LayerOutput:
        return LayerOutput(self)

    @property
    def tap(self) -> Optional[str]:
        """The MDC_TAP_* name that yields this layer's output; None = the model's own output (softmax rows)."""
        kind, i = self.model.topology.kind, self.index
        table = {"deployed": {2: "conv", 3: "flat", 4: "dense", 5: None, 6: None},
                 "cnnpy": {2: "conv", 3: "flat", 4: "hidden", 5: "dense", 6: None, 7: None},
                 # Dropout is the identity at inference: its output is its input's
                 "vtcnn2": {5: "conv", 6: "conv", 7: "flat", 8: "hidden", 9: "hidden", 10: "dense", 11: None, 12: None}}[kind]
        if i not in table:
            raise ValueError(f"{self!r}: this layer's output is never materialised by the kernels (input reshapes / paddings; "
                             "VT-CNN2's conv1 feeds conv2 from registers) -- no tap")
        return table[i]


class SubModel:
    """`Model(inputs=model.inputs, outputs=model.layers[i].output)`: predict() = one forward of the parent with that tap."""

    def __init__(self, parent: VTCNN2, layer: Layer):
        self.parent, self.layer, self._tap = parent, layer, layer.tap      # (raises here for a layer without a tap)

    def compile(self, *_a, **_k) -> None:
        """The notebook compiles its sub-models before predicting; nothing to prepare here."""

    @property
    def output_shape(self):
        return self.layer.output_shape

    def predict(self, X, batch_size: Optional[int] = None, verbose: int = 0):
        out = self.parent.predict(X, batch_size=batch_size, tap=self._tap)
        return out.reshape((out.shape[0],) + tuple(self.layer.output_shape[1:]))


def Model(inputs=None, outputs=None) -> SubModel:
    """keras.models.Model(inputs=model.inputs, outputs=model.layers[i].output) for a layer of a VTCNN2 (CNN.ipynb cell 17)."""
    if not isinstance(outputs, LayerOutput):
        raise TypeError("outputs must be `model.layers[i].output` of a VTCNN2")
    parent = outputs.layer.model
    if inputs is not None and not (isinstance(inputs, ModelInputs) and inputs.model is parent):
        raise ValueError("inputs must be the same model's `model.inputs`")
    return SubModel(parent, outputs.layer)
