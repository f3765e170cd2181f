"""CPU oracle of the reference's TRAINING step (numpy).  TEST INFRASTRUCTURE ONLY -- same rule as oracle_np.py: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this; the product never does.

What it restates (file:line into /root/reference):

* ``model.compile(loss='categorical_crossentropy', optimizer='adam')``  cnn.py:113, CNN.ipynb cell 6
* ``model.fit(X_train, Y_train, batch_size=1024, epochs=100, validation_data=(X_test, Y_test), callbacks=[
  ModelCheckpoint(filepath, monitor='val_loss', save_best_only=True), EarlyStopping(monitor='val_loss', patience=5)])``
  then ``model.load_weights(filepath)``                                    cnn.py:122-147, CNN.ipynb cell 7-8
  for the two nets the reference trains: the deployed net of CNN.ipynb cell 6 (T1 / T2) and cnn.py:104-112's literal
  model (T4).  Neither has a Dropout layer (``dr`` at cnn.py:102 / CNN.ipynb cell 6 is never used), so training and
  inference forward passes are the same function.

The arithmetic lives in third-party Keras 2.4.0 / TensorFlow 2.4.0 (versions from the .h5 attributes), absent here.
Their published semantics, restated:

* loss (tf.keras.backend.categorical_crossentropy, from_logits=False -- the model ends in Activation('softmax') followed
  by Reshape, so the loss sees PROBABILITIES, not a Softmax op whose logits it could reuse):
      q = p / sum(p);  q = clip(q, 1e-7, 1 - 1e-7);  L_i = -sum_c y_ic log q_ic;  loss = mean_i L_i
  (clip_by_value passes the gradient where min <= q <= max, blocks it outside).
* ReLU gradient: passes where the pre-activation is > 0 (ReluGrad).
* Adam (optimizer_v2/adam.py -> training_ops ApplyAdam, non-Nesterov, no amsgrad), t = iterations + 1, all f32:
      alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
      m += (g - m) * (1 - beta1);  v += (g*g - v) * (1 - beta2);  var -= (m * alpha) / (sqrt(v) + eps)
  with lr = 1e-3, beta1 = 0.9, beta2 = 0.999, eps = 1e-7 (the values stored in every bundled .h5's training_config).
* EarlyStopping(monitor='val_loss', patience=5, min_delta=0): best = inf; after each epoch: if val < best: best = val,
  wait = 0, else wait += 1 and stop once wait >= patience.  ModelCheckpoint(save_best_only=True): save when val < best.
  fit() shuffles the training set every epoch (here: the caller supplies the permutations) and keeps the last, short batch.

PINNING: **parity unpinned** -- the dataset is not bundled and the reference records no training run that could be
replayed (loss curves are figures; CNN.ipynb cell 9's 0.5455 needs the dataset).  tests/test_oracle_train.py holds this
file against torch-CPU autograd of the same loss (gradients) and against torch.optim.Adam at eps = 0 plus a second,
independent restatement of the update (Adam).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

Weights = List[Tuple[np.ndarray, np.ndarray]]

KERAS_EPSILON = 1e-7          # K.epsilon(): the clip of categorical_crossentropy
ADAM_DEFAULTS = dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7)      # Keras 2.4 Adam() defaults = the .h5 training_config


def _as(x, dtype):
    return np.ascontiguousarray(np.asarray(x), dtype=dtype)


def _softmax(z):
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def crossentropy_on_probs(p: np.ndarray, y: np.ndarray):
    """Per-sample Keras categorical cross-entropy on probability rows and its gradient w.r.t. those rows."""
    dt = p.dtype.type
    eps = dt(KERAS_EPSILON)
    s = p.sum(axis=-1, keepdims=True)
    q = p / s
    qc = np.clip(q, eps, dt(1) - eps)
    loss = -(y * np.log(qc)).sum(axis=-1)
    passes = (q >= eps) & (q <= dt(1) - eps)
    gq = np.where(passes, -y / qc, dt(0))
    gp = (gq - (gq * q).sum(axis=-1, keepdims=True)) / s
    return loss, gp


def _softmax_backward(p, gp):
    return p * (gp - (gp * p).sum(axis=-1, keepdims=True))


# ---------------------------------------------------------------------------------------------------------------------
# Optional Dropout(dr) behind the conv activations (and behind cnn.py's Dense(10)) -- the positions the reference's `dr` was
# written for: the DeepSig notebook it derives from has `model.add(Dropout(dr))` after every conv and after dense1
# (RML2016.10a_VTCNN2_example.ipynb:229-243); the reference's own definitions (cnn.py:104-112, CNN.ipynb cell 6) dropped those
# layers and kept the variable.  Keras: training output = x * mask / (1 - rate), mask ~ Bernoulli(1 - rate) per element and
# step; inference: identity.  TensorFlow's generator cannot be replayed, so the library uses a STATED counter-based one
# (include/mdc.h, mdc_trainer_set_dropout), restated here bit for bit:
#     fmix32(h):  h ^= h >> 16;  h *= 0x85EBCA6B;  h ^= h >> 13;  h *= 0xC2B2AE35;  h ^= h >> 16        (murmur3's finaliser)
#     k_step  = fmix32(seed + 0x9E3779B9 * (step + 1))            step = Adam's iteration count before this update
#     k_frame = fmix32(k_step ^ (frame * 0x85EBCA6B + site))      frame = the frame's index in the data set, site 0 conv / 1 dense1
#     u       = fmix32(k_frame + element * 0xC2B2AE35)            element = index in the layer's Flatten order
#     keep iff u >= floor(rate * 2^32)
# ---------------------------------------------------------------------------------------------------------------------
_M32 = np.uint64(0xFFFFFFFF)


def _fmix32(h):
    h = np.asarray(h, np.uint64) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def dropout_scale(rate: float, seed: int, step: int, frames, nelem: int, site: int, dtype=np.float64) -> np.ndarray:
    """(len(frames), nelem) array of 0 or 1 / (1 - rate): what the kernels multiply the layer's output by."""
    frames = np.asarray(frames, np.uint64) & _M32
    k_step = _fmix32((np.uint64(seed) + np.uint64(0x9E3779B9) * np.uint64(step + 1)) & _M32)
    k_frame = _fmix32(k_step ^ ((frames * np.uint64(0x85EBCA6B) + np.uint64(site)) & _M32))
    u = _fmix32((k_frame[:, None] + np.arange(nelem, dtype=np.uint64)[None, :] * np.uint64(0xC2B2AE35)) & _M32)
    thr = np.uint64(int(np.floor(np.float64(np.float32(rate)) * 4294967296.0)))
    return np.where(u >= thr, dtype(1) / (dtype(1) - dtype(np.float32(rate))), dtype(0)).astype(dtype)


def loss_and_grads_deployed(x, y, weights: Weights, dtype=np.float64, dropout: Optional[dict] = None):
    """CNN.ipynb cell 6 net: returns (mean loss, per-sample losses, [(dkernel, dbias)] of the MEAN loss, probs).
    dropout = {'rate', 'seed', 'step', 'frames'}: the optional Dropout behind the conv activations (see above)."""
    (ck, cb), (wd, bd) = weights
    x, y = _as(x, dtype), _as(y, dtype)
    F = ck.shape[-1]
    k = _as(ck, dtype).reshape(2, F)
    b, wd, bd = _as(cb, dtype), _as(wd, dtype), _as(bd, dtype)
    n = x.shape[0]
    xp = np.zeros((n, 2, 130), dtype)
    xp[:, :, 1:129] = x
    x0, x1 = xp[:, :, 0:129, None], xp[:, :, 1:130, None]
    pre = x0 * k[0] + x1 * k[1] + b
    a = np.maximum(pre, 0)
    flat = a.reshape(n, 258 * F)
    keep = None
    if dropout is not None and dropout["rate"] > 0:
        keep = dropout_scale(dropout["rate"], dropout["seed"], dropout["step"], dropout["frames"], 258 * F, 0, dtype)
        flat = flat * keep
    z = flat @ wd + bd
    d = np.maximum(z, 0)
    p = _softmax(d)
    li, gp = crossentropy_on_probs(p, y)
    gz = _softmax_backward(p, gp) * (z > 0) / dtype(n)
    dwd = flat.T @ gz
    dbd = gz.sum(axis=0)
    gflat = gz @ wd.T
    if keep is not None:
        gflat = gflat * keep
    ga = gflat.reshape(n, 2, 129, F) * (pre > 0)
    dk = np.stack([(ga * x0).sum(axis=(0, 1, 2)), (ga * x1).sum(axis=(0, 1, 2))]).reshape(ck.shape)
    db = ga.sum(axis=(0, 1, 2))
    return float(li.mean(dtype=np.float64)), li, [(dk, db), (dwd, dbd)], p


def loss_and_grads_cnnpy(x, y, weights: Weights, dtype=np.float64, dropout: Optional[dict] = None):
    """cnn.py:104-112 as TensorFlow builds it: (H, W, C) = (1, 2, 128), pad W by 1, Conv2D(F,(1,2)) -> (1,3,F).
    dropout: optional Dropout behind the conv activations (site 0) and behind Dense(D, relu) (site 1)."""
    (ck, cb), (w1, b1), (w2, b2) = weights
    x, y = _as(x, dtype), _as(y, dtype)
    F = ck.shape[-1]
    k = _as(ck, dtype).reshape(2, 128, F)
    cb, w1, b1, w2, b2 = (_as(t, dtype) for t in (cb, w1, b1, w2, b2))
    n = x.shape[0]
    xp = np.zeros((n, 4, 128), dtype)
    xp[:, 1:3, :] = x
    pre = np.stack([xp[:, w] @ k[0] + xp[:, w + 1] @ k[1] for w in range(3)], axis=1) + cb      # (n,3,F)
    a = np.maximum(pre, 0)
    flat = a.reshape(n, 3 * F)
    keep0 = keep1 = None
    if dropout is not None and dropout["rate"] > 0:
        keep0 = dropout_scale(dropout["rate"], dropout["seed"], dropout["step"], dropout["frames"], 3 * F, 0, dtype)
        keep1 = dropout_scale(dropout["rate"], dropout["seed"], dropout["step"], dropout["frames"], w1.shape[1], 1, dtype)
        flat = flat * keep0
    z1 = flat @ w1 + b1
    h = np.maximum(z1, 0)
    if keep1 is not None:
        h = h * keep1
    lg = h @ w2 + b2
    p = _softmax(lg)
    li, gp = crossentropy_on_probs(p, y)
    glg = _softmax_backward(p, gp) / dtype(n)
    dw2, db2 = h.T @ glg, glg.sum(axis=0)
    gh = glg @ w2.T
    if keep1 is not None:
        gh = gh * keep1
    gz1 = gh * (z1 > 0)
    dw1, db1 = flat.T @ gz1, gz1.sum(axis=0)
    gflat = gz1 @ w1.T
    if keep0 is not None:
        gflat = gflat * keep0
    ga = gflat.reshape(n, 3, F) * (pre > 0)
    dk = np.zeros((2, 128, F), dtype)
    for w in range(3):
        dk[0] += xp[:, w].T @ ga[:, w]
        dk[1] += xp[:, w + 1].T @ ga[:, w]
    return float(li.mean(dtype=np.float64)), li, [(dk.reshape(ck.shape), ga.sum(axis=(0, 1))), (dw1, db1), (dw2, db2)], p


def loss_and_grads(kind: str, x, y, weights: Weights, dtype=np.float64, dropout: Optional[dict] = None):
    if kind == "deployed":
        return loss_and_grads_deployed(x, y, weights, dtype, dropout)
    if kind == "cnnpy":
        return loss_and_grads_cnnpy(x, y, weights, dtype, dropout)
    raise ValueError(f"the reference trains the deployed and cnn.py nets only, not {kind!r}")


def onehot(labels, classes: int, dtype=np.float32) -> np.ndarray:
    """to_onehot of cnn.py:74-81."""
    lab = np.asarray(labels).astype(np.int64)
    out = np.zeros((len(lab), classes), dtype)
    out[np.arange(len(lab)), lab] = 1
    return out


class KerasAdam:
    """TensorFlow 2.4's ApplyAdam on a list of float32 tensors (see the module docstring)."""

    def __init__(self, shapes: Sequence[Tuple[int, ...]], lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
        f = np.float32
        self.lr, self.beta1, self.beta2, self.eps = f(lr), f(beta1), f(beta2), f(eps)
        self.m = [np.zeros(s, np.float32) for s in shapes]
        self.v = [np.zeros(s, np.float32) for s in shapes]
        self.iterations = 0

    def apply(self, params: List[np.ndarray], grads: List[np.ndarray]) -> None:
        f = np.float32
        t = self.iterations + 1
        b1p, b2p = f(np.power(self.beta1, f(t))), f(np.power(self.beta2, f(t)))
        alpha = f(self.lr * np.sqrt(f(1) - b2p) / (f(1) - b1p))
        for p, g, m, v in zip(params, grads, self.m, self.v):
            g = g.astype(np.float32)
            m += (g - m) * (f(1) - self.beta1)
            v += (g * g - v) * (f(1) - self.beta2)
            p -= (m * alpha) / (np.sqrt(v) + self.eps)
        self.iterations = t


def flatten_weights(weights: Weights) -> List[np.ndarray]:
    return [t for pair in weights for t in pair]


def train_step(kind: str, x, y, weights: Weights, opt: KerasAdam, dtype=np.float32, dropout: Optional[dict] = None) -> float:
    """train_on_batch: gradients of the batch's mean loss in `dtype`, then the Adam update of the f32 weights IN PLACE.
    dropout = {'rate', 'seed', 'frames'}: the step is the optimizer's iteration count."""
    if dropout is not None:
        dropout = dict(dropout, step=opt.iterations)
    loss, _li, grads, _p = loss_and_grads(kind, x, y, weights, dtype, dropout)
    opt.apply(flatten_weights(weights), flatten_weights(grads))
    return loss


def evaluate(kind: str, x, y, weights: Weights, dtype=np.float32, chunk: int = 4096) -> float:
    """model.evaluate: mean per-sample loss over the whole set."""
    tot, n = 0.0, len(x)
    for s in range(0, n, chunk):
        _l, li, _g, _p = loss_and_grads(kind, x[s:s + chunk], y[s:s + chunk], weights, dtype)
        tot += float(li.sum(dtype=np.float64))
    return tot / max(n, 1)


def fit(kind: str, weights: Weights, x, y, batch_size: int, epochs: int, validation_data, patience: Optional[int] = 5,
        permutations: Optional[Callable[[int], np.ndarray]] = None, dtype=np.float32, adam: Optional[dict] = None) -> Dict:
    """The loop of cnn.py:122-147.  `permutations(epoch)` supplies the epoch's shuffle (Keras' own is unseeded).
    Returns {'loss': [...], 'val_loss': [...], 'best_epoch', 'stopped_epoch', 'best_weights', 'weights', 'opt'}."""
    weights = [(k.astype(np.float32).copy(), b.astype(np.float32).copy()) for k, b in weights]
    opt = KerasAdam([t.shape for t in flatten_weights(weights)], **(adam or {}))
    xv, yv = validation_data
    n = len(x)
    hist = {"loss": [], "val_loss": []}
    best, wait, best_epoch, stopped, best_w = np.inf, 0, -1, None, None
    for ep in range(epochs):
        order = permutations(ep) if permutations is not None else np.arange(n)
        tot = 0.0
        for s in range(0, n, batch_size):
            idx = order[s:s + batch_size]
            tot += train_step(kind, x[idx], y[idx], weights, opt, dtype) * len(idx)
        hist["loss"].append(tot / n)
        val = evaluate(kind, xv, yv, weights, dtype)
        hist["val_loss"].append(val)
        if val < best:                      # ModelCheckpoint(save_best_only) and EarlyStopping share the comparison
            best, wait, best_epoch = val, 0, ep
            best_w = [(k.copy(), b.copy()) for k, b in weights]
        else:
            wait += 1
            if patience is not None and wait >= patience:
                stopped = ep
                break
    hist.update(best_epoch=best_epoch, stopped_epoch=stopped, best_weights=best_w, weights=weights, opt=opt)
    return hist
