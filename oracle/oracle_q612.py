"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the FPGA's Q6.12 integer arithmetic for the deployed nets.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path is
modulationdetectioncnn_amd/csrc/deployed_q612.hip behind mdc_forward_q612.

What it follows (cnn_test_latest1.sv, read as text; nothing here is simulated RTL):
  * operands are 18-bit two's-complement Q6.12 words (value = int / 4096), as produced by `float2fix`
    (CNN.ipynb cell 23: truncate |v| * 2**12 toward 0) and stored in the ROM tables (sv:685-707, 719-3132);
  * conv neuron (signed_mult1, sv:642-658):  mult_out[35:0] = a*b + c*d;  out = {mult_out[35], mult_out[28:12]};
    out_add = out + e (18-bit wrap);  relu = out_add[17] ? 0 : out_add.   a,c = the two samples of the (1,2) kernel,
    b,d = its two taps, e = bias.  Zero padding of one sample on each side (conv_layer, sv:476-481);
  * dense term (signed_mult, sv:664-675): the same bit selection of I_act*W_i + Q_act*W_q, sign-extended and added
    into a 32-bit accumulator that starts from the sign-extended 18-bit bias (dense_layer, sv:293-343);
  * final ReLU on the 32-bit sums (layers_top, sv:171-176).
The ORDER of activations/weights is Keras' (CNN.ipynb cell 6 / SURVEY.md 8(a) A1, A6): the RTL's clocking, ROM
latency and its reversed write index (sv:507) are not modelled -- this is an arithmetic model, not a simulation.

Parity: the arithmetic rules above are read off the SystemVerilog; no RTL output was ever recorded in the
reference and no simulator exists in the image, so bit-level agreement with the FPGA is **parity unpinned**.  What
IS pinned (tests/test_q612.py): the results agree with the float oracle on the reference's recorded vectors to the
quantisation error (CNN.ipynb cell 18 / 12.16.testDataYunyun.txt frame 0), and the frozen labels are reproduced.
"""
from __future__ import annotations

import numpy as np

FRAC = 12
WIDTH = 18


def wrap(v, bits):
    """Two's-complement wrap of integer array v to `bits` bits."""
    v = np.asarray(v, dtype=np.int64)
    m = np.int64(1) << (bits - 1)
    return ((v + m) & ((np.int64(1) << bits) - 1)) - m


def quantize(x) -> np.ndarray:
    """float2fix (CNN.ipynb cell 23): truncate |v|*4096 toward zero, 18-bit two's complement (-0 repaired to 0)."""
    x = np.asarray(x, dtype=np.float64)
    return wrap(np.trunc(x * (1 << FRAC)).astype(np.int64), WIDTH)


def select18(acc36):
    """{mult_out[35], mult_out[28:12]} of a 36-bit signed sum of two products (sv:655, 674), as a signed value."""
    acc36 = np.asarray(acc36, dtype=np.int64)
    low = (acc36 >> FRAC) & ((1 << 17) - 1)
    # mult_out is a 36-bit wire: its sign is bit 35 of the WRAPPED sum (a*b + c*d = +2^35, reached only with all
    # four operands at -2^17, wraps to -2^35 and selects a negative value)
    return low - (((acc36 >> 35) & 1) << 17)


def quantize_weights(weights):
    """[(conv kernel (1,2,1,F), bias F), (dense kernel (258F, C), bias C)] floats -> the same shapes as Q6.12 ints."""
    (ck, cb), (dk, db) = weights
    return [(quantize(ck), quantize(cb)), (quantize(dk), quantize(db))]


def forward_q612(xq, wq):
    """xq (n,2,128) Q6.12 ints; wq from quantize_weights.  Returns conv (n,2,129,F), dense (n,C) int32-valued
    (post-ReLU, value = int/4096), labels (n) = first maximum (np.argmax, cnn.py:209)."""
    xq = wrap(np.asarray(xq, dtype=np.int64), WIDTH)
    n = xq.shape[0]
    (ck, cb), (dk, db) = wq
    F = ck.shape[-1]
    k0 = ck.reshape(2, F)[0].astype(np.int64)
    k1 = ck.reshape(2, F)[1].astype(np.int64)
    xp = np.zeros((n, 2, 130), dtype=np.int64)
    xp[:, :, 1:129] = xq
    a = xp[:, :, :129, None]          # x[h][w-1]
    c = xp[:, :, 1:, None]            # x[h][w]
    out = select18(a * k0 + c * k1)                      # (n,2,129,F)
    out_add = wrap(out + cb.astype(np.int64), WIDTH)
    conv = np.where(out_add >= 0, out_add, 0)
    flat = conv.reshape(n, 2, 129 * F)                   # Keras Flatten of (2,129,F): h, then w*F + f
    C = dk.shape[1]
    wi = dk[:129 * F].astype(np.int64)                   # rows of the I half, (129F, C)
    wqd = dk[129 * F:].astype(np.int64)
    term = select18(flat[:, 0, :, None] * wi[None] + flat[:, 1, :, None] * wqd[None])     # (n,129F,C)
    acc = wrap(term.sum(axis=1) + db.astype(np.int64)[None], 32)
    dense = np.where(acc >= 0, acc, 0)
    labels = np.argmax(dense, axis=1).astype(np.int32) if n else np.zeros((0,), np.int32)
    return {"conv": conv, "dense": dense.astype(np.int64).reshape(n, C), "labels": labels}


def forward_from_float(x, weights):
    """Quantise float frames and float weights the way the reference's table writer does, then run the integer net."""
    return forward_q612(quantize(x), quantize_weights(weights))
