"""TEST INFRASTRUCTURE ONLY -- the same CPU restatement as oracle_np.py, written with torch-CPU library ops
(F.conv2d, matmul, softmax) so that the host cores get a competitive multithreaded baseline
("time the build's torch-CPU restatement").  Only tests/ and bench.py's cpu_baseline leg import it.

It follows the same reference lines as oracle_np.py: CNN.ipynb cell 6 (deployed), cnn.py:104-115 (cnnpy),
RML2016.10a_VTCNN2_example.ipynb:229-243 (vtcnn2); Keras semantics = cross-correlation, 'valid', zero padding on W
only, Flatten in the layer's data format.  tests/test_oracle_crosscheck.py holds it against oracle_np (which is the
one pinned to the reference's golden vectors)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))


def forward(kind: str, x, weights):
    """x (n,2,128) float32; weights as the Keras get_weights() pairs.  Returns probs (n,C) and first-max labels."""
    x = _t(x)
    n = x.shape[0]
    with torch.no_grad():
        if kind == "deployed":
            (ck, cb), (dk, db) = weights
            Fn = ck.shape[-1]
            w = _t(ck).reshape(2, Fn).t().reshape(Fn, 1, 1, 2)                   # HWIO (1,2,1,F) -> OIHW
            y = F.relu(F.conv2d(F.pad(x[:, None], (1, 1)), w, _t(cb)))            # (n,F,2,129)
            flat = y.permute(0, 2, 3, 1).reshape(n, 258 * Fn)                     # channels_last Flatten: h, w, f
            logits = F.relu(flat @ _t(dk) + _t(db))
        elif kind == "vtcnn2":
            (k1, b1), (k2, b2), (d1, c1), (d2, c2) = weights
            y = F.relu(F.conv2d(F.pad(x[:, None], (2, 2)), _t(k1), _t(b1)))       # (n,256,2,130), OIHW kernels
            y = F.relu(F.conv2d(F.pad(y, (2, 2)), _t(k2), _t(b2)))                # (n,80,1,132)
            h = F.relu(y.reshape(n, 80 * 132) @ _t(d1) + _t(c1))                  # channels_first Flatten: c*132 + w
            logits = h @ _t(d2) + _t(c2)
        elif kind == "cnnpy":
            (ck, cb), (d1, c1), (d2, c2) = weights
            Fn = ck.shape[-1]
            xc = x.permute(0, 2, 1)[:, :, None, :]                                # (n,128,1,2): H=1, W=2, C=128 as NCHW
            w = _t(ck).permute(3, 2, 0, 1)                                        # HWIO (1,2,128,F) -> OIHW
            y = F.relu(F.conv2d(F.pad(xc, (1, 1)), w, _t(cb)))                    # (n,F,1,3)
            flat = y.permute(0, 2, 3, 1).reshape(n, 3 * Fn)
            logits = F.relu(flat @ _t(d1) + _t(c1)) @ _t(d2) + _t(c2)
        else:
            raise ValueError(kind)
        probs = torch.softmax(logits, dim=1)
    p = probs.numpy()
    return {"probs": p, "labels": np.argmax(p, axis=1).astype(np.int32) if n else np.zeros((0,), np.int32)}
