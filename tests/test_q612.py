"""The Q6.12 integer path.

CPU (`not gpu`): the integer oracle (oracle/oracle_q612.py) against what the reference recorded -- the float
results of CNN.ipynb cell 18 / 12.16.testDataYunyun.txt (to the quantisation error) and the frozen labels.
GPU: mdc_forward_q612 == the integer oracle, bit for bit, including 18-bit / 32-bit wrap-around.
Bit-level agreement with the FPGA itself is parity-unpinned: the reference holds no RTL outputs."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_deployed_npz
from oracle import oracle_np as O
from oracle import oracle_q612 as Q


def _txt_weights(name):
    z = np.load(os.path.join(GOLDEN, "weights_txt", name + ".npz"))
    return [(z["conv_kernel"], z["conv_bias"]), (z["dense_kernel"], z["dense_bias"])]


def _frames():
    raw = np.load(os.path.join(GOLDEN, "frames.npz"))["raw"]
    meta = json.load(open(os.path.join(GOLDEN, "frames.json")))
    return raw.astype(np.int64), meta


def test_bit_selection_and_wrap_rules():
    assert Q.select18(np.array([4096])) == 1 and Q.select18(np.array([4095])) == 0
    assert Q.select18(np.array([-1])) == -1 and Q.select18(np.array([-4096])) == -1 and Q.select18(np.array([-4097])) == -2
    # {m[35], m[28:12]} drops bits 34..29: a sum of 2^29 selects to 0, of -2^29 to -2^17 + 0
    assert Q.select18(np.array([1 << 29])) == 0
    assert Q.select18(np.array([-(1 << 29)])) == -(1 << 17)
    assert Q.wrap(np.array([1 << 17]), 18) == -(1 << 17)
    assert list(Q.quantize(np.array([0.5, -0.5, -1e-5, 1e-5, 31.9999, -32.0]))) == [2048, -2048, 0, 0, 131071, -131072]


def _verilog_select18(a, b, c, d):
    """signed_mult / signed_mult1 of cnn_test_latest1.sv:642-675 restated on Python integers, independently of the
    oracle: `wire signed [35:0] mult_out = a*b + c*d` (36-bit wrap), `out = {mult_out[35], mult_out[28:12]}`."""
    m = (a * b + c * d) & ((1 << 36) - 1)
    out = (((m >> 35) & 1) << 17) | ((m >> 12) & 0x1FFFF)
    return out - (1 << 18) if out & (1 << 17) else out


def test_select18_takes_its_sign_from_bit_35_of_the_36_bit_wire():
    lo, hi = -(1 << 17), (1 << 17) - 1
    corners = [lo, lo + 1, -1, 0, 1, hi - 1, hi]
    quads = [(a, b, c, d) for a in corners for b in corners for c in corners for d in corners]
    rng = np.random.default_rng(3)
    quads += [tuple(int(v) for v in rng.integers(lo, hi + 1, 4)) for _ in range(2000)]
    for a, b, c, d in quads:
        assert int(Q.select18(np.array([a * b + c * d]))[0]) == _verilog_select18(a, b, c, d), (a, b, c, d)
    # the one case where the unwrapped sum's sign differs from bit 35: all four operands -2^17 -> +2^35 -> wraps
    assert int(Q.select18(np.array([1 << 35]))[0]) == -(1 << 17) == _verilog_select18(lo, lo, lo, lo)


def test_matches_recorded_keras_outputs_to_quantisation_error():
    """12.16.testDataYunyun.txt frame 0 is already Q6.12; with the 12.15.latestWeights.txt tables (the SV ROM) the
    integer net must land on Keras' recorded [0, 3.1391976, 0.3649335] within the truncation error."""
    raw, meta = _frames()
    i0 = meta["names"].index("12.16.testDataYunyun.txt#0")
    wq = Q.quantize_weights(_txt_weights("12.15.latestWeights"))
    r = Q.forward_q612(raw[i0:i0 + 1], wq)
    got = r["dense"][0] / 4096.0
    assert np.abs(got - np.array(meta["keras_prediction"][i0])).max() < 0.25      # 774 truncated terms of <= 2^-12 each
    assert r["labels"][0] == 1


def test_reproduces_the_frozen_labels_on_every_bundled_frame():
    """Same decision as the float oracle wherever that decision is not a near-tie."""
    raw, meta = _frames()
    w = load_deployed_npz("3convmodrecnets_CNN2_0.5")
    flt = O.forward_deployed(raw.astype(np.float64) / 4096.0, *[a for p in w for a in p], dtype=np.float64)
    r = Q.forward_q612(raw, Q.quantize_weights(w))
    srt = np.sort(flt["dense"], axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 0.5
    assert decided.sum() >= 10
    assert (r["labels"][decided] == flt["labels"][decided]).all()
    assert np.abs(r["dense"] / 4096.0 - flt["dense"]).max() < 0.5


def test_txt_tables_are_exact_in_q612():
    """Weights that came from a .txt table are multiples of 2^-12: quantising them is the identity."""
    w = _txt_weights("12.15.latestWeights")
    for (k, b), (kq, bq) in zip(w, Q.quantize_weights(w)):
        np.testing.assert_array_equal(kq, np.round(k * 4096).astype(np.int64))
        np.testing.assert_array_equal(bq, np.round(b * 4096).astype(np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("name,n", [("3convmodrecnets_CNN2_0.5", 1), ("3convmodrecnets_CNN2_0.5", 257),
                                    ("convmodrecnets_CNN2_0.5", 100), ("3convmodrecnets_CNN2_0.5", 0)])
def test_gpu_q612_is_bit_exact(name, n):
    import torch
    from modulationdetectioncnn_amd import VTCNN2, Topology
    w = load_deployed_npz(name)
    m = VTCNN2(Topology.deployed(w[0][1].shape[0], 3))
    m.set_weights(w)
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((n, 2, 128)) * 0.3).astype(np.float32)
    if n > 4:
        x[1] *= 200.0          # saturating frame: exercises the 18-bit and 32-bit wrap-around
        x[2, 0, 5] = -1e-5     # float2fix "-0"
    ref = Q.forward_from_float(x, w)
    dense, labels = m.predict_q612(x, as_float=False)
    np.testing.assert_array_equal(dense.astype(np.int64), ref["dense"].reshape(n, 3))
    np.testing.assert_array_equal(labels, ref["labels"])
    # integer input takes the same path
    d2, l2 = m.predict_q612(torch.from_numpy(Q.quantize(x).astype(np.int32)), as_float=False)
    np.testing.assert_array_equal(d2.cpu().numpy(), dense)
    np.testing.assert_array_equal(l2.cpu().numpy(), labels)


@pytest.mark.gpu
@pytest.mark.parametrize("F", [3, 10])
def test_gpu_q612_all_operands_at_minus_2_17(F):
    """a*b + c*d = +2^35 (samples and taps all -32.0 = -2^17) wraps to -2^35 on the 36-bit wire: the conv neuron
    selects -2^17, the bias -1 wraps the 18-bit sum to +131071, and ReLU lets it through (a sign taken from the
    unwrapped 64-bit sum would give 0 - 1 -> ReLU 0 instead)."""
    from modulationdetectioncnn_amd import VTCNN2, Topology
    rng = np.random.default_rng(11)
    ck = np.full((1, 2, 1, F), -32.0, np.float32)
    cb = np.full((F,), -1.0 / 4096.0, np.float32)
    dk = (rng.integers(-2048, 2048, (258 * F, 3)) / 4096.0).astype(np.float32)
    db = (rng.integers(-2048, 2048, (3,)) / 4096.0).astype(np.float32)
    w = [(ck, cb), (dk, db)]
    x = (rng.standard_normal((70, 2, 128)) * 0.3).astype(np.float32)
    x[0] = -32.0
    x[5, 0, :] = -32.0
    x[9, 1, 40:90] = -32.0
    ref = Q.forward_from_float(x, w)
    assert (ref["conv"][0, :, 1:128, :] == 131071).all()          # interior positions of the all -32.0 frame
    m = VTCNN2(Topology.deployed(F, 3))
    m.set_weights(w)
    dense, labels = m.predict_q612(x, as_float=False)
    np.testing.assert_array_equal(dense.astype(np.int64), ref["dense"])
    np.testing.assert_array_equal(labels, ref["labels"])


@pytest.mark.gpu
def test_gpu_q612_bundled_frames_and_float_view():
    from modulationdetectioncnn_amd import VTCNN2
    raw, meta = _frames()
    m = VTCNN2.from_npz(os.path.join(GOLDEN, "weights_txt", "12.15.latestWeights.npz"))
    dense, labels = m.predict_q612(raw.astype(np.int32))
    ref = Q.forward_q612(raw, Q.quantize_weights(_txt_weights("12.15.latestWeights")))
    np.testing.assert_array_equal(labels, ref["labels"])
    np.testing.assert_array_equal(dense, (ref["dense"] / 4096.0).astype(np.float32))
    with pytest.raises(ValueError):
        VTCNN2.synthetic("vtcnn2", classes=3).predict_q612(np.zeros((1, 2, 128), np.float32))
