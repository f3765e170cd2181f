"""Device-side confusion counts (cnn.py:205-216) and the uint8 I/Q front-end."""
import numpy as np
import pytest
import torch

from conftest import load_deployed_npz
from modulationdetectioncnn_amd import VTCNN2, Topology, frames_from_iq_u8, synthetic_frames
from modulationdetectioncnn_amd.sharding import confusion_counts

pytestmark = pytest.mark.gpu


def _t1():
    w = load_deployed_npz("3convmodrecnets_CNN2_0.5")
    m = VTCNN2(Topology.deployed(3, 3))
    m.set_weights(w)
    return m


@pytest.mark.parametrize("n", [0, 1, 1000, 100003])
def test_confusion_counts_match_the_reference_loop(n):
    m = _t1()
    x = synthetic_frames(n, seed=3, device="cuda") * 4.0
    truth = np.random.default_rng(1).integers(0, 3, size=n).astype(np.int32)
    counts = m.confusion_counts_device(x, truth).cpu().numpy()
    pred = m.predict_classes(x).cpu().numpy()
    want = np.zeros((3, 3), np.int64)
    for j, k in zip(truth, pred):          # cnn.py:205-210, literally
        want[j, k] += 1
    np.testing.assert_array_equal(counts, want)
    np.testing.assert_array_equal(counts, confusion_counts(truth, pred, 3, reduce=False))
    assert counts.sum() == n
    if n:
        conf = m.confusion(x, truth)
        rows = want.sum(axis=1, keepdims=True)
        np.testing.assert_allclose(conf, np.divide(want, rows, out=np.zeros((3, 3)), where=rows > 0))
        assert m.accuracy(x, truth) == pytest.approx(np.trace(want) / n)


def test_confusion_rejects_labels_out_of_range():
    m = _t1()
    x = synthetic_frames(8, seed=3, device="cuda")
    with pytest.raises(ValueError):
        m.confusion_counts_device(x, np.array([0, 1, 2, 3, 0, 0, 0, 0]))
    with pytest.raises(ValueError):
        m.confusion_counts_device(x, np.zeros(7, np.int32))


@pytest.mark.parametrize("n", [0, 1, 33, 4097])
def test_iq_u8_frontend_is_bit_exact(n):
    rng = np.random.default_rng(9)
    iq = rng.integers(0, 256, size=256 * n, dtype=np.uint8)
    x = frames_from_iq_u8(iq)
    assert x.shape == (n, 2, 128) and x.dtype == torch.float32
    pairs = iq.reshape(n, 128, 2).astype(np.float32)
    scale = np.float32(1.0 / 127.5)
    want = np.stack([(pairs[:, :, 0] - np.float32(127.5)) * scale, (pairs[:, :, 1] - np.float32(127.5)) * scale], axis=1)
    np.testing.assert_array_equal(x.cpu().numpy(), want)
    if n:
        # straight into the classifier, device to device
        m = _t1()
        p = m.predict(x * 0.02)
        assert p.shape == (n, 3) and torch.isfinite(p).all()


def _iq_to_frames_np(iq, scale):
    """The conversion of mdc_iq_u8_to_frames restated in numpy float32 (I row 0, Q row 1)."""
    pairs = iq.reshape(-1, 128, 2).astype(np.float32)
    sc = np.float32(scale)
    return np.stack([(pairs[:, :, 0] - np.float32(127.5)) * sc, (pairs[:, :, 1] - np.float32(127.5)) * sc], axis=1)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("name", ["3convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5"])      # T1 (F=3), T2 (F=10)
@pytest.mark.parametrize("n", [0, 1, 15, 63, 64, 65, 191, 4096, 70001])
def test_fused_raw_iq_forward_equals_the_two_pass_path_bit_for_bit(name, n, dtype):
    """mdc_forward_iq_u8 (bytes read by the forward kernel) == mdc_iq_u8_to_frames + mdc_forward, exactly; and both
    agree with the f64 oracle run on the numpy restatement of the conversion."""
    import os
    from conftest import GOLDEN
    from oracle import oracle_np as O
    m = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"), dtype=dtype)
    rng = np.random.default_rng(100 + n)
    iq = rng.integers(0, 256, size=256 * n, dtype=np.uint8)
    scale = 0.02 / 127.5                         # samples of the size of the reference's frames
    t = torch.from_numpy(iq).cuda()
    probs, labels = m.predict_iq_u8(t, scale)
    assert probs.shape == (n, 3) and labels.shape == (n,) and labels.dtype == torch.int32
    x = frames_from_iq_u8(t, scale)
    p2, l2, _ = m.forward_device(x)
    assert torch.equal(probs, p2) and torch.equal(labels, l2)
    if 0 < n <= 4096:
        w = [a for p in load_deployed_npz(name) for a in p]
        ref = O.forward_deployed(_iq_to_frames_np(iq, scale), *w, dtype=np.float64)
        np.testing.assert_allclose(probs.cpu().numpy(), ref["probs"], atol=2e-6 if dtype == "f32" else 1e-2)
    # numpy in -> numpy out
    if n == 65:
        pn, ln = m.predict_iq_u8(iq, scale)
        assert isinstance(pn, np.ndarray) and np.array_equal(pn, probs.cpu().numpy()) and np.array_equal(ln, labels.cpu().numpy())


def test_fused_raw_iq_extreme_bytes_and_default_scale():
    """All-0, all-255 and alternating bytes (the sample range ends) through the fused kernel, default scale 1/127.5."""
    m = _t1()
    iq = np.concatenate([np.zeros(256 * 64, np.uint8), np.full(256 * 64, 255, np.uint8),
                         np.tile(np.array([0, 255], np.uint8), 128 * 64), np.tile(np.array([255, 0], np.uint8), 128 * 3)])
    probs, labels = m.predict_iq_u8(torch.from_numpy(iq).cuda())
    x = frames_from_iq_u8(iq)
    p2, l2, _ = m.forward_device(x)
    assert torch.isfinite(probs).all() and torch.equal(probs, p2) and torch.equal(labels, l2)
    assert float(x.min()) == -1.0 and float(x.max()) == 1.0


def test_raw_iq_other_topologies_and_rejects():
    """VT-CNN2 takes raw bytes through the device-side conversion (same call); the fused C entry refuses it."""
    from modulationdetectioncnn_amd import _cabi
    iq = np.random.default_rng(5).integers(0, 256, size=256 * 40, dtype=np.uint8)
    m = VTCNN2.synthetic(Topology.vtcnn2(11), seed=2016, device=0, dtype="f32")
    probs, labels = m.predict_iq_u8(iq, 0.02 / 127.5)
    want = m.predict(frames_from_iq_u8(iq, 0.02 / 127.5))
    np.testing.assert_array_equal(probs, want.cpu().numpy())
    t = torch.from_numpy(iq).cuda()
    out = torch.empty((40, 11), dtype=torch.float32, device="cuda")
    rc = _cabi.lib().mdc_forward_iq_u8(m._engine(), t.data_ptr(), 40, 1.0, out.data_ptr(), None, None)
    assert rc == -95 and b"deployed" in _cabi.lib().mdc_last_error()
    d = _t1()
    with pytest.raises(ValueError):
        d.predict_iq_u8(np.zeros(300, np.uint8))
    with pytest.raises(TypeError):
        d.predict_iq_u8(torch.zeros(256, dtype=torch.int16))
    with pytest.raises(_cabi.MdcError):          # misaligned byte pointer
        _cabi.check(_cabi.lib().mdc_forward_iq_u8(d._engine(), t.data_ptr() + 4, 1, 1.0, None, None, None))


def test_iq_u8_rejects_partial_frames():
    with pytest.raises(ValueError):
        frames_from_iq_u8(np.zeros(300, np.uint8))
    with pytest.raises(TypeError):
        frames_from_iq_u8(torch.zeros(256, dtype=torch.int16))


def test_accuracy_by_snr_matches_the_reference_loop():
    """cnn.py:228-259 literally, against VTCNN2.accuracy_by_snr."""
    m = _t1()
    n = 5000
    x = synthetic_frames(n, seed=4, device="cuda") * 4.0
    rng = np.random.default_rng(2)
    truth = rng.integers(0, 3, size=n)
    snrs = rng.choice([-20, -10, 0, 10, 18], size=n)
    acc, conf = m.accuracy_by_snr(x, truth, snrs)
    pred = m.predict_classes(x).cpu().numpy()
    for snr in sorted(set(snrs.tolist())):
        sel = np.where(snrs == snr)[0]
        want = np.zeros((3, 3))
        for i in sel:                                   # cnn.py:242-245
            want[truth[i], pred[i]] += 1
        cor = np.sum(np.diag(want)); ncor = np.sum(want) - cor
        np.testing.assert_array_equal(conf[snr], want.astype(np.int64))
        assert acc[snr] == pytest.approx(1.0 * cor / (cor + ncor))
    assert set(acc) == set(snrs.tolist())
