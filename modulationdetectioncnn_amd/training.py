"""Host-side mirror of the reference's training calls, over libmdc.so's mdc_trainer_* entry points.

    reference (cnn.py / CNN.ipynb)                                        here
    -------------------------------------------------------------------   ------------------------------------------
    model.compile(loss='categorical_crossentropy', optimizer='adam')      Trainer(topology, weights)        cnn.py:113
    history = model.fit(X_train, Y_train, batch_size=1024, epochs=100,    history = m.fit(X_train, Y_train, batch_size=1024,
        validation_data=(X_test, Y_test), callbacks=[                         epochs=100, validation_data=(X_test, Y_test),
        ModelCheckpoint(filepath, monitor='val_loss', save_best_only=True),   checkpoint=filepath,
        EarlyStopping(monitor='val_loss', patience=5)])                       patience=5)                   cnn.py:135-146
    model.load_weights(filepath)                                          m.load_weights(filepath)          cnn.py:147
    history.epoch, history.history['loss'], ['val_loss']                  the same attributes               cnn.py:164-166

Scope: the two nets the reference trains -- Topology.deployed (CNN.ipynb cell 6) and Topology.cnnpy (cnn.py:104-112) --
in f32.  All arithmetic (forward, loss, backward, Adam) runs in hand-written gfx950 kernels (csrc/train.hip); there is no
CPU fallback.  The training set is placed in HBM once; an epoch's shuffle is an index array, a mini-batch two launches,
and the host reads one pair of numbers (loss, val_loss) per epoch.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _cabi
from .topology import Topology

_KIND = {"deployed": _cabi.KIND_DEPLOYED, "vtcnn2": _cabi.KIND_VTCNN2, "cnnpy": _cabi.KIND_CNNPY}
ADAM_DEFAULTS = dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7)      # keras.optimizers.Adam(); every bundled .h5's training_config

Weights = List[Tuple[np.ndarray, np.ndarray]]


def _torch():
    import torch
    return torch


def to_onehot(labels, classes: Optional[int] = None) -> np.ndarray:
    """cnn.py:74-81 (`yy1[np.arange(len(data)), data] = 1`), float32."""
    lab = np.asarray(labels).astype(np.int64).reshape(-1)
    classes = int(lab.max()) + 1 if classes is None else int(classes)
    if lab.size and (lab.min() < 0 or lab.max() >= classes):
        raise ValueError(f"labels lie outside [0, {classes})")
    out = np.zeros((lab.size, classes), np.float32)
    out[np.arange(lab.size), lab] = 1.0
    return out


class History:
    """What Keras' fit returns, as far as the reference reads it (cnn.py:164-166): .epoch and .history[...]."""

    def __init__(self):
        self.epoch: List[int] = []
        self.history: Dict[str, List[float]] = {"loss": [], "val_loss": []}
        self.best_epoch: Optional[int] = None
        self.stopped_epoch: Optional[int] = None      # EarlyStopping.stopped_epoch (None: ran all epochs)
        self.best_weights: Optional[Weights] = None

    def __repr__(self):
        return f"History(epochs={len(self.epoch)}, best_epoch={self.best_epoch}, stopped_epoch={self.stopped_epoch})"


class Trainer:
    """The compiled model's training state on one MI355X: f32 master weights, Adam moments, step count."""

    def __init__(self, topology: Topology, weights: Sequence[Tuple[np.ndarray, np.ndarray]], device: Optional[int] = None,
                 lr: float = 1e-3, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-7, dropout: float = 0.0,
                 dropout_seed: int = 0, _lib_variant: str = "product"):
        """dropout: OPTIONAL Dropout(rate) behind the conv activations (and cnn.py's Dense(10)) in training batches -- 0, the
        default, is the reference's nets, which contain no Dropout layer (mdc_trainer_set_dropout in include/mdc.h states the
        counter-based mask generator)."""
        if topology.kind not in ("deployed", "cnnpy"):
            raise ValueError("the reference trains the deployed (CNN.ipynb cell 6) and cnn.py (cnn.py:104-112) nets; "
                             f"training is not built for {topology.kind!r}")
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device visible: the MI355X training path has no CPU fallback")
        self.topology = topology
        self._variant = _lib_variant
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self._h = C.c_void_p()
        L = self._lib()
        topo = _cabi.MdcTopology(_KIND[topology.kind], topology.filters, topology.hidden, topology.classes, (C.c_int32 * 4)(0, 0, 0, 0))
        self._check(L.mdc_trainer_create(C.byref(topo), self.device_index, C.byref(self._h)))
        self.adam = dict(lr=float(lr), beta1=float(beta1), beta2=float(beta2), eps=float(eps))
        self._check(L.mdc_trainer_set_adam(self._h, *[self.adam[k] for k in ("lr", "beta1", "beta2", "eps")]))
        self.dropout, self.dropout_seed = float(dropout), int(dropout_seed) & 0xFFFFFFFF
        if self.dropout:
            self._check(L.mdc_trainer_set_dropout(self._h, self.dropout, self.dropout_seed))
        self._set(_cabi.TRAIN_WEIGHTS, weights)

    # ------------------------------------------------------------------ plumbing
    def _lib(self):
        return _cabi.lib(self._variant)

    def _check(self, rc: int) -> int:
        return _cabi.check(rc, self._variant)

    def _stream(self):
        torch = _torch()
        return torch.cuda.current_stream(torch.device("cuda", self.device_index)).cuda_stream

    def close(self) -> None:
        if self._h is not None and self._h.value:
            self._lib().mdc_trainer_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _set(self, which: int, tensors: Sequence[Tuple[np.ndarray, np.ndarray]]) -> None:
        shapes = self.topology.layer_shapes
        if len(tensors) != len(shapes):
            raise ValueError(f"expected {len(shapes)} (kernel, bias) pairs, got {len(tensors)}")
        fp = C.POINTER(C.c_float)
        for i, ((k, b), (ks, bs)) in enumerate(zip(tensors, shapes)):
            k = np.ascontiguousarray(k, dtype=np.float32)
            b = np.ascontiguousarray(b, dtype=np.float32)
            if tuple(k.shape) != ks or tuple(b.shape) != bs:
                raise ValueError(f"layer {i}: expected kernel {ks} bias {bs}, got {k.shape} {b.shape}")
            self._check(self._lib().mdc_trainer_set_tensor(self._h, which, i, k.ctypes.data_as(fp), k.size, b.ctypes.data_as(fp), b.size,
                                                           self._stream()))

    def _get(self, which: int) -> Weights:
        fp = C.POINTER(C.c_float)
        out = []
        for i, (ks, bs) in enumerate(self.topology.layer_shapes):
            k, b = np.empty(ks, np.float32), np.empty(bs, np.float32)
            self._check(self._lib().mdc_trainer_get_tensor(self._h, which, i, k.ctypes.data_as(fp), k.size, b.ctypes.data_as(fp), b.size,
                                                           self._stream()))
            out.append((k, b))
        return out

    # ------------------------------------------------------------------ state
    def get_weights(self) -> Weights:
        return self._get(_cabi.TRAIN_WEIGHTS)

    def set_weights(self, weights) -> None:
        self._set(_cabi.TRAIN_WEIGHTS, weights)

    def gradients(self) -> Weights:
        """d(mean loss)/d(weights) of the last train_batch, in the weights' layouts."""
        return self._get(_cabi.TRAIN_GRADIENT)

    def optimizer_state(self) -> Dict:
        """What the /optimizer_weights group of a Keras full-model .h5 holds: Adam's iter, m and v per tensor."""
        return {"iterations": self.read(reset=False)["iterations"], "m": self._get(_cabi.TRAIN_ADAM_M), "v": self._get(_cabi.TRAIN_ADAM_V)}

    def set_optimizer_state(self, state: Dict) -> None:
        self._set(_cabi.TRAIN_ADAM_M, state["m"])
        self._set(_cabi.TRAIN_ADAM_V, state["v"])
        self._check(self._lib().mdc_trainer_set_iterations(self._h, int(state["iterations"]), self._stream()))

    # ------------------------------------------------------------------ data
    def _frames(self, X):
        torch = _torch()
        dev = torch.device("cuda", self.device_index)
        x = X if isinstance(X, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(X), dtype=np.float32))
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        if x.ndim != 3 or tuple(x.shape[1:]) != (2, 128):
            raise ValueError(f"expected frames of shape (n,2,128); got {tuple(x.shape)}")
        return x

    def _targets(self, Y, n: int):
        torch = _torch()
        dev = torch.device("cuda", self.device_index)
        Cn = self.topology.classes
        y = Y if isinstance(Y, torch.Tensor) else torch.from_numpy(np.asarray(Y))
        if y.ndim == 1:                         # class indices -> the one-hot rows of cnn.py:74-82
            if y.numel() and (int(y.min()) < 0 or int(y.max()) >= Cn):
                raise ValueError(f"labels lie outside [0, {Cn})")
            y = torch.nn.functional.one_hot(y.to(torch.int64), Cn)
        y = y.to(device=dev, dtype=torch.float32).contiguous()
        if tuple(y.shape) != (n, Cn):
            raise ValueError(f"expected targets of shape ({n},{Cn}) or ({n},); got {tuple(y.shape)}")
        return y

    # ------------------------------------------------------------------ steps
    def _resident(self, x, y, order=None) -> None:
        """The raw pointers below are dereferenced by the kernels: refuse anything that is not a contiguous tensor of the expected
        type in THIS device's memory (a host tensor's data_ptr() would be a GPU fault, not an error).  The VALUES of `order` are
        checked where they are read, on the device: a position that names no frame of x is skipped and the next read() raises
        (fit() also checks the permutations it is given on the host, before anything is enqueued)."""
        torch = _torch()
        want = (("frames", x, torch.float32, (2, 128)), ("targets", y, torch.float32, (self.topology.classes,)))
        for name, t, dt, tail in want + ((("order", order, torch.int32, ()),) if order is not None else ()):
            if not isinstance(t, torch.Tensor) or not t.is_cuda or t.device.index != self.device_index or t.dtype != dt \
                    or not t.is_contiguous() or tuple(t.shape[1:]) != tail:
                raise ValueError(f"{name}: expected a contiguous {dt} tensor (n,{','.join(map(str, tail))}) on cuda:{self.device_index}")
        if y.shape[0] != x.shape[0]:
            raise ValueError(f"{x.shape[0]} frames but {y.shape[0]} target rows")

    def train_batch(self, x, y, order=None, first: int = 0, count: Optional[int] = None, apply: bool = True) -> None:
        """model.train_on_batch on frames order[first:first+count] of the device-resident set (x, y); enqueue only."""
        self._resident(x, y, order)
        n = x.shape[0] if order is None else order.shape[0]
        count = n - first if count is None else int(count)
        if first < 0 or count < 0 or first + count > n:
            raise ValueError(f"batch [{first}, {first + count}) outside the {n} frames")
        self._check(self._lib().mdc_train_batch(self._h, x.data_ptr(), y.data_ptr(), x.shape[0], order.data_ptr() if order is not None else None,
                                                first, count, int(apply), self._stream()))

    def evaluate_enqueue(self, x, y) -> None:
        self._resident(x, y)
        self._check(self._lib().mdc_trainer_evaluate(self._h, x.data_ptr(), y.data_ptr(), x.shape[0], None, 0, x.shape[0], self._stream()))

    def read(self, reset: bool = True) -> Dict:
        tl, el = C.c_double(), C.c_double()
        tf, ef, it = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib().mdc_trainer_read(self._h, int(reset), C.byref(tl), C.byref(tf), C.byref(el), C.byref(ef), C.byref(it),
                                                 self._stream()))
        return {"train_loss_sum": tl.value, "train_frames": tf.value, "eval_loss_sum": el.value, "eval_frames": ef.value,
                "iterations": it.value}

    def evaluate(self, X, Y) -> float:
        """model.evaluate(X, Y): the mean categorical cross-entropy with the weights as they are now."""
        x = self._frames(X)
        y = self._targets(Y, x.shape[0])
        self.read(reset=True)
        self.evaluate_enqueue(x, y)
        r = self.read(reset=True)
        return r["eval_loss_sum"] / max(r["eval_frames"], 1)

    def loss_and_gradients(self, X, Y) -> Tuple[float, Weights]:
        """Mean loss of one batch and its gradient, nothing applied (the test hook for the backward pass)."""
        x = self._frames(X)
        y = self._targets(Y, x.shape[0])
        self.read(reset=True)
        self.train_batch(x, y, apply=False)
        r = self.read(reset=True)
        return r["train_loss_sum"] / max(r["train_frames"], 1), self.gradients()

    # ------------------------------------------------------------------ the loop of cnn.py:135-146
    def fit(self, X, Y, batch_size: int = 1024, epochs: int = 100, validation_data=None, patience: Optional[int] = 5,
            checkpoint: Optional[str] = None, shuffle: bool = True, seed: Optional[int] = None, permutations=None,
            verbose: int = 0, on_best=None, callbacks=None) -> History:
        """model.fit(X, Y, batch_size, epochs, validation_data=(Xv, Yv), callbacks=[ModelCheckpoint(checkpoint,
        monitor='val_loss', save_best_only=True), EarlyStopping(monitor='val_loss', patience=patience)]) -- pass the two
        callbacks themselves (callbacks.py: cnn.py:143-144 verbatim) or their short forms `checkpoint=` / `patience=`.
        Every epoch trains on a fresh shuffle (numpy Generator(seed); Keras' own shuffle is unseeded) in batches of
        batch_size -- the last one short --, then computes val_loss; an improvement (strictly smaller) saves the
        checkpoint and resets the patience counter, `patience` epochs without one stop the run.  Without
        validation_data a callback that monitors val_loss watches nothing (Keras warns and skips it): all epochs run,
        nothing is saved.  `permutations(epoch)` overrides the shuffle (tests).  on_best(epoch, val_loss, trainer): called
        whenever val_loss improves on every earlier epoch."""
        from .callbacks import EarlyStopping, ModelCheckpoint
        if callbacks is not None:
            if checkpoint is not None:
                raise ValueError("pass callbacks=[ModelCheckpoint(...)] or checkpoint=..., not both")
            ckpts = [c for c in callbacks if isinstance(c, ModelCheckpoint)]
            stops = [c for c in callbacks if isinstance(c, EarlyStopping)]
            if len(ckpts) + len(stops) != len(callbacks) or len(ckpts) > 1 or len(stops) > 1:
                raise ValueError("callbacks: at most one ModelCheckpoint and one EarlyStopping of this package's callbacks module")
            ckpt, stop = (ckpts or [None])[0], (stops or [None])[0]
        else:
            ckpt = ModelCheckpoint(checkpoint, save_best_only=True) if checkpoint is not None else None
            stop = EarlyStopping(patience=int(patience)) if patience is not None else None
        for c in (ckpt, stop):
            if c is not None:
                c.reset()
        torch = _torch()
        x = self._frames(X)
        n = x.shape[0]
        y = self._targets(Y, n)
        if n == 0:
            raise ValueError("fit needs at least one frame")
        if batch_size < 1:
            raise ValueError("batch_size must be >= 1")
        xv = yv = None
        if validation_data is not None:
            xv = self._frames(validation_data[0])
            yv = self._targets(validation_data[1], xv.shape[0])
        rng = np.random.default_rng(seed)
        hist = History()
        best, restore = np.inf, None
        dev = torch.device("cuda", self.device_index)
        self.read(reset=True)
        for ep in range(int(epochs)):
            if permutations is not None:
                order_h = np.asarray(permutations(ep), dtype=np.int32)
            else:
                order_h = (rng.permutation(n) if shuffle else np.arange(n)).astype(np.int32)
            if order_h.shape != (n,) or (n and (int(order_h.min()) < 0 or int(order_h.max()) >= n)):
                raise ValueError(f"a permutation needs one index in [0, {n}) per frame")
            order = torch.from_numpy(order_h).to(dev)
            for s in range(0, n, batch_size):
                self.train_batch(x, y, order, s, min(batch_size, n - s), apply=True)
            if xv is not None:
                self.evaluate_enqueue(xv, yv)
            r = self.read(reset=True)                       # the epoch's one synchronisation
            loss = r["train_loss_sum"] / max(r["train_frames"], 1)
            hist.epoch.append(ep)
            hist.history["loss"].append(loss)
            val = r["eval_loss_sum"] / max(r["eval_frames"], 1) if xv is not None else None
            logs = {"loss": loss, "val_loss": val}
            if val is not None:
                hist.history["val_loss"].append(val)
            if verbose:
                print(f"Epoch {ep + 1}/{epochs} - loss: {loss:.4f}" + (f" - val_loss: {val:.4f}" if val is not None else ""))
            if val is not None and val < best:
                best, hist.best_epoch = val, ep
                hist.best_weights = self.get_weights()
                if on_best is not None:
                    on_best(ep, val, self)
            if ckpt is not None and ckpt.should_save(logs[ckpt.monitor]):
                self.save(ckpt.filepath)
            if stop is not None:
                improved, halt = stop.update(logs[stop.monitor])
                if improved and stop.restore_best_weights:
                    restore = self.get_weights()
                if halt:
                    hist.stopped_epoch = stop.stopped_epoch = ep
                    if stop.restore_best_weights and restore is not None:
                        self.set_weights(restore)
                    break
        if xv is None:
            hist.history.pop("val_loss")
        return hist

    def save(self, filepath: str) -> None:
        """model.save(filepath) as ModelCheckpoint calls it (cnn.py:143): a Keras 2.4 full-model HDF5 file -- topology JSON,
        weights, training_config and Adam's state -- that the reference's model.load_weights(filepath) accepts."""
        from .formats.h5mini import write_keras_h5
        write_keras_h5(filepath, self.topology, self.get_weights(), optimizer=self.optimizer_state(), adam=self.adam)
