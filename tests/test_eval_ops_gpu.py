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


def _windows_np(iq, hop, scale):
    """mdc_iq_u8_windows restated in numpy float32: window i = pairs [i*hop, i*hop + 128)."""
    pairs = iq.reshape(-1, 2)
    n = 0 if len(pairs) < 128 else (len(pairs) - 128) // hop + 1
    idx = (np.arange(n)[:, None] * hop + np.arange(128)[None, :])
    w = pairs[idx].astype(np.float32)                    # (n,128,2)
    sc = np.float32(scale)
    return np.stack([(w[:, :, 0] - np.float32(127.5)) * sc, (w[:, :, 1] - np.float32(127.5)) * sc], axis=1)


@pytest.mark.parametrize("hop", [128, 64, 7, 1])
def test_iq_u8_windows_are_bit_exact(hop):
    rng = np.random.default_rng(hop)
    iq = rng.integers(0, 256, size=2 * (128 + hop * 300), dtype=np.uint8)
    x = frames_from_iq_u8(iq, 0.01, hop=hop)
    want = _windows_np(iq, hop, 0.01)
    assert x.shape == want.shape == (301, 2, 128)
    np.testing.assert_array_equal(x.cpu().numpy(), want)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("name", ["3convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5"])
@pytest.mark.parametrize("hop", [128, 64, 8, 3, 1])
def test_sliding_windows_fused_equals_convert_then_forward(name, dtype, hop):
    """A live capture classified every `hop` sample pairs (README.md:5): the forward kernel reading the overlapping
    windows straight from the byte stream == mdc_iq_u8_windows + mdc_forward, bit for bit (odd hops leave the windows
    2-byte aligned only)."""
    import os
    from conftest import GOLDEN
    m = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"), dtype=dtype)
    for n in (1, 64, 65, 1000, 4097):
        rng = np.random.default_rng(1000 * hop + n)
        iq = rng.integers(0, 256, size=2 * (128 + hop * (n - 1)), dtype=np.uint8)
        t = torch.from_numpy(iq).cuda()
        scale = 0.02 / 127.5
        probs, labels = m.predict_iq_u8(t, scale, hop=hop)
        assert probs.shape == (n, 3)
        p2, l2, _ = m.forward_device(frames_from_iq_u8(t, scale, hop=hop))
        assert torch.equal(probs, p2) and torch.equal(labels, l2), (hop, n)
        # chunked (chunk boundaries at arbitrary windows): same bits
        p3, l3 = m.predict_iq_u8(t, scale, hop=hop, batch_size=97)
        assert torch.equal(probs, p3) and torch.equal(labels, l3)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "fp8"])
@pytest.mark.parametrize("hop", [128, 64, 5, 1])
def test_vtcnn2_reads_raw_bytes_in_its_conv_staging(dtype, hop):
    """(f)3 for the canonical VT-CNN2: the conv kernels' frame staging converts the uint8 pairs itself; logits-level
    outputs (probabilities, labels) are bit-identical to convert-then-forward."""
    m = VTCNN2.synthetic(Topology.vtcnn2(11), seed=2016, device=0, dtype=dtype)
    scale = 0.02 / 127.5
    for n in (1, 16, 17, 100, 700):
        rng = np.random.default_rng(77 * hop + n)
        iq = rng.integers(0, 256, size=2 * (128 + hop * (n - 1)), dtype=np.uint8)
        t = torch.from_numpy(iq).cuda()
        probs, labels = m.predict_iq_u8(t, scale, hop=hop)
        p2, l2, _ = m.forward_device(frames_from_iq_u8(t, scale, hop=hop))
        assert probs.shape == (n, 11) and torch.equal(probs, p2) and torch.equal(labels, l2), (dtype, hop, n)
    p3, l3 = m.predict_iq_u8(t, scale, hop=hop, batch_size=48)
    assert torch.equal(p3, p2) and torch.equal(l3, l2)


def test_raw_iq_other_topologies_and_rejects():
    """cnn.py's literal model takes raw bytes through the device-side conversion (same Python call); the fused C entry
    refuses it; bad arguments are errors, not crashes."""
    from modulationdetectioncnn_amd import _cabi
    iq = np.random.default_rng(5).integers(0, 256, size=256 * 40, dtype=np.uint8)
    m = VTCNN2.synthetic("cnnpy", classes=5, seed=2016, device=0)
    probs, labels = m.predict_iq_u8(iq, 0.02 / 127.5)
    want = m.predict(frames_from_iq_u8(iq, 0.02 / 127.5))
    np.testing.assert_array_equal(probs, want.cpu().numpy())
    t = torch.from_numpy(iq).cuda()
    out = torch.empty((40, 5), dtype=torch.float32, device="cuda")
    L = _cabi.lib()
    rc = L.mdc_forward_iq_u8(m._engine(), t.data_ptr(), 40, 128, 1.0, out.data_ptr(), None, None, 0, None)
    assert rc == -95 and b"deployed" in L.mdc_last_error()
    d = _t1()
    with pytest.raises(ValueError):
        d.predict_iq_u8(np.zeros(300, np.uint8))
    with pytest.raises(ValueError):
        d.predict_iq_u8(np.zeros(301, np.uint8), hop=5)          # half a pair
    with pytest.raises(TypeError):
        d.predict_iq_u8(torch.zeros(256, dtype=torch.int16))
    with pytest.raises(_cabi.MdcError):          # odd byte pointer: not a whole (I,Q) pair
        _cabi.check(L.mdc_forward_iq_u8(d._engine(), t.data_ptr() + 1, 1, 128, 1.0, None, None, None, 0, None))
    with pytest.raises(_cabi.MdcError):          # hop < 1
        _cabi.check(L.mdc_forward_iq_u8(d._engine(), t.data_ptr(), 1, 0, 1.0, None, None, None, 0, None))
    v = VTCNN2.synthetic(Topology.vtcnn2(3), seed=1, device=0, dtype="bf16")
    with pytest.raises(_cabi.MdcError):          # vtcnn2 without its workspace
        _cabi.check(L.mdc_forward_iq_u8(v._engine(), t.data_ptr(), 16, 128, 1.0, None, None, None, 0, None))
    assert d.predict_iq_u8(np.zeros(100, np.uint8), hop=3)[0].shape == (0, 3)      # capture shorter than one window


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


@pytest.mark.parametrize("classes,bins,n", [(3, 5, 5000), (11, 20, 100003), (32, 40, 20000), (2, 1, 10)])
def test_binned_confusion_one_launch_matches_the_literal_loop(classes, bins, n):
    """mdc_confusion_binned (LDS histogram, or global atomics when bins*C*C exceeds the LDS table: 40 x 32 x 32)."""
    from modulationdetectioncnn_amd import _cabi
    rng = np.random.default_rng(classes * 100 + bins)
    truth = rng.integers(0, classes, size=n).astype(np.int32)
    pred = rng.integers(0, classes, size=n).astype(np.int32)
    b = rng.integers(0, bins, size=n).astype(np.int32)
    truth[3] = classes          # out of range -> bad
    b[7] = -1                   # out of range -> bad
    want = np.zeros((bins, classes, classes), np.int64)
    for i in range(n):          # cnn.py:242-245 per SNR bin
        if i not in (3, 7):
            want[b[i], truth[i], pred[i]] += 1
    tt, pp, bb = (torch.from_numpy(a).cuda() for a in (truth, pred, b))
    counts = torch.zeros((bins, classes, classes), dtype=torch.int64, device="cuda")
    bad = torch.zeros((1,), dtype=torch.int64, device="cuda")
    _cabi.check(_cabi.lib().mdc_confusion_binned(tt.data_ptr(), pp.data_ptr(), bb.data_ptr(), n, classes, bins, counts.data_ptr(),
                                                 bad.data_ptr(), torch.cuda.current_stream().cuda_stream))
    np.testing.assert_array_equal(counts.cpu().numpy(), want)
    assert int(bad.item()) == 2


def test_save_results_writes_the_tuple_the_reference_pickles(tmp_path):
    """cnn.py:262-264: cPickle.dump(("CNN2", 0.5, acc), fd) -- acc = {snr: accuracy} from the per-SNR loop."""
    import pickle
    m = _t1()
    n = 3000
    x = synthetic_frames(n, seed=4, device="cuda") * 4.0
    rng = np.random.default_rng(2)
    truth = rng.integers(0, 3, size=n)
    snrs = rng.choice(np.arange(-20, 20, 2), size=n)
    acc, _ = m.accuracy_by_snr(x, truth, snrs)
    path = str(tmp_path / "results_cnn2_d0.5.dat")
    VTCNN2.save_results(path, acc)
    with open(path, "rb") as fd:
        tag, dr, got = pickle.load(fd)                      # a file this test just wrote
    assert (tag, dr) == ("CNN2", 0.5) and got == {int(k): v for k, v in acc.items()}
    assert all(type(k) is int and type(v) is float for k, v in got.items())
    assert VTCNN2.load_results(path) == (tag, dr, got)
    assert open(path, "rb").read(2) == b"\x80\x02"          # protocol 2: readable by the reference's Python-2 cPickle


def test_one_process_multi_stream_driver_on_one_gpu():
    """MultiStreamPredictor (BASELINE configs[3]: per-GPU HIP streams) with G = 1 device and two streams: results equal
    the single-stream forward, bit for bit, for every topology family."""
    from modulationdetectioncnn_amd.sharding import MultiStreamPredictor
    for m, n in ((_t1(), 10001), (VTCNN2.synthetic(Topology.vtcnn2(11), seed=2016, device=0, dtype="bf16"), 5000)):
        msp = MultiStreamPredictor.for_models([m], streams_per_device=2)
        assert len(msp.lanes) == 2 and msp.lanes[0].stream != msp.lanes[1].stream
        x = synthetic_frames(n, seed=9, device="cuda")
        ref_p, ref_l, _ = m.forward_device(x)
        torch.cuda.synchronize()
        plan = msp.plan(n)
        outs = msp.forward_shards([x[lo:hi].contiguous() for _, lo, hi in plan])
        assert torch.equal(torch.cat([o[0] for o in outs]), ref_p) and torch.equal(torch.cat([o[1] for o in outs]), ref_l)
        p, l = msp.predict(x.cpu().numpy())
        np.testing.assert_array_equal(p, ref_p.cpu().numpy())
        np.testing.assert_array_equal(l, ref_l.cpu().numpy())


@pytest.mark.parametrize("kind,n", [("t1", 1), ("t1", 4097), ("t1", 200003), ("vtcnn2", 3000)])
def test_evaluate_is_keras_categorical_crossentropy(kind, n):
    """`score = model.evaluate(X_test, Y_test)` (cnn.py:153): the mean categorical cross-entropy of the softmax rows, Keras'
    way (row / sum, clip to [1e-7, 1 - 1e-7], -log of the true entry) -- mdc_crossentropy against the numpy restatement on
    the SAME probabilities, and one-hot targets (cnn.py:80-82) == index targets.  Parity unpinned against the recorded
    0.5455 of CNN.ipynb cell 9 (the dataset it was computed on is not available)."""
    from oracle import oracle_np as O
    if kind == "t1":
        m, C = _t1(), 3
        x = synthetic_frames(n, seed=13, device="cuda") * 6.0          # all three classes occur, some rows saturate
    else:
        m, C = VTCNN2.synthetic(Topology.vtcnn2(11), seed=2016, dtype="bf16"), 11
        x = synthetic_frames(n, seed=13, device="cuda")
    truth = np.random.default_rng(2).integers(0, C, size=n).astype(np.int32)
    p = m.predict(x).cpu().numpy()
    want = O.categorical_crossentropy(p, truth)
    got = m.evaluate(x, truth)
    assert got == pytest.approx(want, rel=2e-6)
    onehot = np.zeros((n, C))
    onehot[np.arange(n), truth] = 1
    assert m.evaluate(x.cpu().numpy(), onehot, batch_size=1024) == pytest.approx(want, rel=2e-6)
    # the clip: a row that puts (almost) nothing on the true class costs -log(1e-7), not infinity
    hard = p.argmin(axis=1).astype(np.int32)
    capped = m.evaluate(x, hard)
    assert np.isfinite(capped) and capped <= -np.log(1e-7) * (1 + 1e-6)
    assert capped == pytest.approx(O.categorical_crossentropy(p, hard), rel=2e-6)
    with pytest.raises(ValueError):
        m.evaluate(x, np.full(n, C, np.int32))


def test_crossentropy_entry_point_accumulates_and_validates():
    from modulationdetectioncnn_amd import _cabi
    L = _cabi.lib()
    p = torch.tensor([[0.5, 0.25, 0.25], [0.1, 0.8, 0.1]], dtype=torch.float32, device="cuda")
    t = torch.tensor([0, 1], dtype=torch.int32, device="cuda")
    acc = torch.zeros(1, dtype=torch.float64, device="cuda")
    for _ in range(3):                                                  # caller-zeroed, accumulates across calls (shards)
        _cabi.check(L.mdc_crossentropy(p.data_ptr(), t.data_ptr(), 2, 3, acc.data_ptr(), None, None))
    torch.cuda.synchronize()
    assert float(acc.item()) == pytest.approx(3 * (-np.log(0.5) - np.log(0.8)), rel=1e-6)
    L.mdc_last_error.restype = __import__("ctypes").c_char_p
    assert L.mdc_crossentropy(p.data_ptr(), t.data_ptr(), -1, 3, acc.data_ptr(), None, None) == -22
    assert L.mdc_crossentropy(None, t.data_ptr(), 2, 3, acc.data_ptr(), None, None) == -22
    assert L.mdc_crossentropy(p.data_ptr(), t.data_ptr(), 2, 3, None, None, None) == -22
    assert L.mdc_crossentropy(p.data_ptr(), t.data_ptr(), 2, 99, acc.data_ptr(), None, None) == -22 and b"classes" in L.mdc_last_error()
    assert L.mdc_crossentropy(None, None, 0, 3, acc.data_ptr(), None, None) == 0


@pytest.mark.filterwarnings("ignore:invalid value encountered")      # the oracle's 0/0 on the all-NaN row is the point of the test
def test_crossentropy_propagates_nan_like_keras():
    """a NaN probability row (NaN / Inf samples in, or a row summing to 0) must make the score NaN, as Keras'
    clip_by_value leaves it and as the numpy oracle's np.clip does -- fmaxf alone would turn it into 1e-7 and the mean loss
    into a finite number."""
    from modulationdetectioncnn_amd import _cabi
    from oracle import oracle_np as O
    L = _cabi.lib()
    p = torch.tensor([[0.2, 0.5, 0.3], [float("nan"), 0.5, 0.5], [0.0, 0.0, 0.0]], dtype=torch.float32, device="cuda")
    for rows, nan in (((0,), False), ((0, 1), True), ((0, 2), True)):
        pp = p[list(rows)].contiguous()
        t = torch.zeros(len(rows), dtype=torch.int32, device="cuda")
        acc = torch.zeros(1, dtype=torch.float64, device="cuda")
        _cabi.check(L.mdc_crossentropy(pp.data_ptr(), t.data_ptr(), len(rows), 3, acc.data_ptr(), None, None))
        got = float(acc.item()) / len(rows)
        want = O.categorical_crossentropy(pp.cpu().numpy(), t.cpu().numpy())
        assert np.isnan(got) == nan == bool(np.isnan(want)), (rows, got, want)
        if not nan:
            assert got == pytest.approx(want, rel=1e-6)
    # (Through VTCNN2.evaluate the score of a frame holding a NaN SAMPLE is a property of the forward kernels, not of this entry
    # point: the deployed f32 kernels' clamp-bit ReLU turns a NaN activation into 0 -- include/mdc.h, "documented range".)
