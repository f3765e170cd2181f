"""CPU oracle: numpy restatement of the reference's inference path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the
checker.  The product (``modulationdetectioncnn_amd``) never imports it and has
no CPU fallback.

What it restates (all file:line are into /root/reference):

* ``forward_deployed``  -- the model of CNN.ipynb cell 6 (T1, F=3) and of the
  ``model_config`` attribute inside convmodrecnets_CNN2_0.5.wts.h5 (T2, F=10):
  Reshape(2,128,1) / ZeroPadding2D((0,1)) / Conv2D(F,(1,2),valid,relu) /
  Flatten (channels_last) / Dense(3,relu) / softmax, run by
  ``model.predict`` (cnn.py:198, CNN.ipynb cell 12/18) and tapped per layer as
  CNN.ipynb cell 17 does (model2=dense, model3=flatten, model4=conv, model5=softmax).
* ``forward_vtcnn2``    -- canonical VT-CNN2,
  examples-master/modulation_recognition/RML2016.10a_VTCNN2_example.ipynb:229-243
  (shapes :190-210).  No weights are bundled; conv is taken as Keras-2 style
  cross-correlation over OIHW kernels.
* ``forward_cnnpy``     -- the literal model of cnn.py:104-115 (T4): with the
  TensorFlow backend ``Reshape([1,2,128])`` is H=1, W=2, C=128.
* ``argmax_first``      -- ``int(np.argmax(row))`` of cnn.py:209 (first max wins).

The arithmetic itself lives in TensorFlow 2.4.0 / Keras 2.4.0 (versions from
the .h5 attributes), which are not vendored and not installed here; the Keras
layer semantics restated are: ZeroPadding2D((0,p)) pads W only; Conv2D is
cross-correlation, 'valid', stride 1, HWIO kernel; Flatten is C-order over the
layer's output tensor; Dense is x@W+b; softmax is exp(x-max)/sum.

PINNING: tests/test_oracle_golden.py checks this file against the Keras
outputs recorded in the reference: T1's Dense output (CNN.ipynb cell 18 and
12.16.testDataYunyun.txt:1-2, :263-264) and, for T2 with the bundled 10-filter
checkpoint, 21 entries of its Flatten output (CNN.ipynb cell 19: three input
samples explain all of them; eighteen are predictions, met to the printed
digits).  T2's Dense layer, T3 and T4 have no recorded outputs: for them this
oracle is "parity unpinned" (cross-checked only against an independent
torch-CPU statement in tests/test_oracle_crosscheck.py).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def softmax(z: np.ndarray) -> np.ndarray:
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=-1, keepdims=True)


def argmax_first(p: np.ndarray) -> np.ndarray:
    """Row-wise np.argmax (first maximum), as the python loop at cnn.py:205-211."""
    return np.argmax(p, axis=-1).astype(np.int32)


def _as(x, dtype):
    return np.ascontiguousarray(np.asarray(x), dtype=dtype)


# ---------------------------------------------------------------------------
# T1 / T2: deployed single-conv nets  (CNN.ipynb cell 6)
# ---------------------------------------------------------------------------
def forward_deployed(x, conv_kernel, conv_bias, dense_kernel, dense_bias, dtype=np.float32) -> Dict[str, np.ndarray]:
    """x (N,2,128); conv_kernel HWIO (1,2,1,F); dense_kernel (258F, C).

    Returns taps named after CNN.ipynb cell 17: 'conv' (N,2,129,F) = model4,
    'flat' (N,258F) = model3, 'dense' (N,C) post-ReLU pre-softmax = model2,
    'probs' = model5 / model.predict, 'labels' = first-max argmax.
    """
    x = _as(x, dtype)
    k = _as(conv_kernel, dtype).reshape(2, -1)           # [kw][f]
    b = _as(conv_bias, dtype)
    wd = _as(dense_kernel, dtype)
    bd = _as(dense_bias, dtype)
    n = x.shape[0]
    xp = np.zeros((n, 2, 130), dtype)                    # ZeroPadding2D((0,1)): W only
    xp[:, :, 1:129] = x
    # y[n,h,w,f] = relu(b[f] + K[0,0,0,f]*xp[w] + K[0,1,0,f]*xp[w+1]),  w = 0..128
    y = xp[:, :, 0:129, None] * k[0] + xp[:, :, 1:130, None] * k[1] + b
    y = np.maximum(y, dtype(0))
    flat = y.reshape(n, 258 * k.shape[1])                # channels_last: h*129F + w*F + f
    dense = np.maximum(flat @ wd + bd, dtype(0))         # Dense(C, activation='relu')
    probs = softmax(dense)
    return {"conv": y, "flat": flat, "dense": dense, "probs": probs, "labels": argmax_first(probs)}


# ---------------------------------------------------------------------------
# T3: canonical VT-CNN2  (RML2016.10a_VTCNN2_example.ipynb:229-243)
# ---------------------------------------------------------------------------
def forward_vtcnn2(x, conv1_kernel, conv1_bias, conv2_kernel, conv2_bias,
                   dense1_kernel, dense1_bias, dense2_kernel, dense2_bias,
                   dtype=np.float32, chunk: int = 64, taps: bool = False) -> Dict[str, np.ndarray]:
    """x (N,2,128); conv1 OIHW (256,1,1,3); conv2 OIHW (80,256,2,3);
    dense1 (10560,256) rows in channels_first flatten order o*132+w; dense2 (256,C).
    Dropout layers are identity at inference.
    """
    x = _as(x, dtype)
    k1 = _as(conv1_kernel, dtype).reshape(conv1_kernel.shape[0], 3)        # [c][j]
    b1 = _as(conv1_bias, dtype)
    k2 = _as(conv2_kernel, dtype)                                           # [o][c][h][j]
    co, ci = k2.shape[0], k2.shape[1]
    k2m = np.ascontiguousarray(k2.transpose(1, 2, 3, 0)).reshape(ci * 6, co)   # rows (c,h,j)
    b2 = _as(conv2_bias, dtype)
    w1, bb1 = _as(dense1_kernel, dtype), _as(dense1_bias, dtype)
    w2, bb2 = _as(dense2_kernel, dtype), _as(dense2_bias, dtype)
    n = x.shape[0]
    out: Dict[str, list] = {"flat": [], "dense1": [], "logits": []}
    if taps:
        out["conv1"] = []
        out["conv2"] = []
    for s in range(0, n, chunk):
        xb = x[s:s + chunk]
        m = xb.shape[0]
        xp = np.zeros((m, 2, 132), dtype)                # ZeroPadding2D((0,2))
        xp[:, :, 2:130] = xb
        # conv1 (1x3), 256 filters: y1[m,c,h,w], w = 0..129
        y1 = (k1[None, :, None, None, 0] * xp[:, None, :, 0:130]
              + k1[None, :, None, None, 1] * xp[:, None, :, 1:131]
              + k1[None, :, None, None, 2] * xp[:, None, :, 2:132]
              + b1[None, :, None, None])
        y1 = np.maximum(y1, dtype(0))
        y1p = np.zeros((m, ci, 2, 134), dtype)           # ZeroPadding2D((0,2))
        y1p[:, :, :, 2:132] = y1
        # conv2 (2x3) over 256 channels, 80 filters: im2col rows (c,h,j) -> matmul
        win = np.lib.stride_tricks.sliding_window_view(y1p, 3, axis=3)     # (m,c,2,132,3)
        a = np.ascontiguousarray(win.transpose(0, 3, 1, 2, 4)).reshape(m * 132, ci * 6)
        y2 = np.maximum(a @ k2m + b2, dtype(0)).reshape(m, 132, co)
        y2 = np.ascontiguousarray(y2.transpose(0, 2, 1))                   # (m,80,132) channels_first
        flat = y2.reshape(m, co * 132)                                     # idx o*132 + w
        d1 = np.maximum(flat @ w1 + bb1, dtype(0))
        lg = d1 @ w2 + bb2
        out["flat"].append(flat)
        out["dense1"].append(d1)
        out["logits"].append(lg)
        if taps:
            out["conv1"].append(y1)
            out["conv2"].append(y2)
    res = {k: (np.concatenate(v) if v else np.zeros((0,), dtype)) for k, v in out.items()}
    res["probs"] = softmax(res["logits"]) if n else np.zeros((0, w2.shape[1]), dtype)
    res["labels"] = argmax_first(res["probs"]) if n else np.zeros((0,), np.int32)
    return res


# ---------------------------------------------------------------------------
# T4: the literal cnn.py model  (cnn.py:104-115)
# ---------------------------------------------------------------------------
def forward_cnnpy(x, conv_kernel, conv_bias, dense1_kernel, dense1_bias,
                  dense2_kernel, dense2_bias, dtype=np.float32) -> Dict[str, np.ndarray]:
    """x (N,2,128) -> Reshape([1,2,128]) = (H=1, W=2, C=128) under channels_last;
    ZeroPadding2D((0,1)) -> W=4; Conv2D(10,(1,2)) HWIO (1,2,128,F) -> (1,3,F);
    Flatten (w,f); Dense(10,relu); Dense(5); softmax."""
    x = _as(x, dtype)
    k = _as(conv_kernel, dtype).reshape(2, 128, -1)       # [kw][c][f]
    n = x.shape[0]
    xp = np.zeros((n, 4, 128), dtype)                    # [w][c]
    xp[:, 1:3, :] = x
    y = np.stack([xp[:, w, :] @ k[0] + xp[:, w + 1, :] @ k[1] for w in range(3)], axis=1)
    y = np.maximum(y + _as(conv_bias, dtype), dtype(0))  # (n,3,F)
    flat = y.reshape(n, -1)
    d1 = np.maximum(flat @ _as(dense1_kernel, dtype) + _as(dense1_bias, dtype), dtype(0))
    lg = d1 @ _as(dense2_kernel, dtype) + _as(dense2_bias, dtype)
    probs = softmax(lg)
    return {"conv": y, "flat": flat, "dense1": d1, "logits": lg, "probs": probs, "labels": argmax_first(probs)}


def forward(kind: str, x, weights, dtype=np.float32, **kw) -> Dict[str, np.ndarray]:
    """Dispatch on topology kind ('deployed' | 'vtcnn2' | 'cnnpy'); weights = [(kernel, bias), ...]."""
    flat = [a for pair in weights for a in pair]
    if kind == "deployed":
        return forward_deployed(x, *flat, dtype=dtype)
    if kind == "vtcnn2":
        return forward_vtcnn2(x, *flat, dtype=dtype, **kw)
    if kind == "cnnpy":
        return forward_cnnpy(x, *flat, dtype=dtype)
    raise ValueError(kind)


def categorical_crossentropy(probs: np.ndarray, labels_true: np.ndarray) -> float:
    """``model.evaluate`` of the reference (cnn.py:113 compiles with loss='categorical_crossentropy', no metrics; :153 prints
    the score): Keras 2.4's categorical_crossentropy on PROBABILITIES -- the model ends in Activation('softmax') followed
    by Reshape, so the loss is handed the softmax output, not logits -- i.e. each row divided by its sum, clipped to
    [1e-7, 1 - 1e-7] (K.epsilon()), minus the log of the true class's entry; mean over the samples.  f32 arithmetic per
    row as TensorFlow's, f64 mean."""
    p = np.asarray(probs, np.float32)
    t = np.asarray(labels_true).astype(np.int64)
    p = p / p.sum(axis=-1, keepdims=True, dtype=np.float32)
    p = np.clip(p, np.float32(1e-7), np.float32(1.0) - np.float32(1e-7))
    return float(np.mean(-np.log(p[np.arange(len(t)), t]).astype(np.float64)))


def confusion(labels_true: np.ndarray, labels_pred: np.ndarray, classes: int) -> np.ndarray:
    """cnn.py:199-216: conf[j,k] += 1 then row-normalise (rows with no samples stay 0)."""
    conf = np.zeros((classes, classes), np.float64)
    np.add.at(conf, (labels_true, labels_pred), 1.0)
    s = conf.sum(axis=1, keepdims=True)
    return np.divide(conf, s, out=np.zeros_like(conf), where=s > 0)
