"""Parity of the HIP deployed-net path (T1 F=3, T2 F=10) against the CPU oracle, through the C ABI.

Tolerances: Dense+ReLU tap 5e-6 abs for the Keras known answer;
2e-6 * scale abs for f32 vs the f64 oracle elsewhere; probabilities 2e-6 abs.  Labels must be
bit-exact wherever the f64 oracle's top-2 margin exceeds 1e-5 (relative to the largest
output); exact ties ([0,0,0] after ReLU) must resolve to the FIRST maximum."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, H5_NAMES, load_deployed_npz
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames, synthetic_weights
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


def _frames():
    raw = np.load(os.path.join(GOLDEN, "frames.npz"))["raw"]
    meta = json.load(open(os.path.join(GOLDEN, "frames.json")))
    return raw.astype(np.float32) / np.float32(4096), meta


_VARIANT = "product"      # the f32_mfma fixture switches to the alternates test build (libmdc_alt.so)


def _model(name, **kw):
    return VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"), _lib_variant=_VARIANT, **kw)


def _check_labels(lab, ref64, tol=1e-5):
    d = ref64["dense"]
    srt = np.sort(d, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    scale = np.maximum(np.abs(d).max(axis=1), 1e-30)
    decided = margin > tol * scale
    exact_tie = margin == 0
    assert (lab[decided] == ref64["labels"][decided]).all()
    assert (lab[exact_tie] == ref64["labels"][exact_tie]).all()      # first-max on exact ties
    return int((~decided & ~exact_tie).sum())


def test_keras_known_answer_on_gpu():
    k = json.load(open(os.path.join(GOLDEN, "keras_kat.json")))
    x = np.asarray(k["input"], np.float32).reshape(1, 2, 128)
    m = _model("3convmodrecnets_CNN2_0.5")
    dense = m.predict(x, tap="dense")
    assert np.abs(dense[0] - np.array(k["keras_dense"])).max() < 5e-6
    assert m.predict_classes(x).tolist() == [0]
    np.testing.assert_allclose(m.predict(x)[0], [0.67148, 0.24728, 0.08124], atol=1e-5)


def test_keras_known_answer_t2_flatten_entries_on_gpu():
    """CNN.ipynb cell 19: 21 entries of the 10-filter net's Flatten output as Keras printed them (the bundled convmodrecnets
    checkpoint; tests/test_oracle_golden.py derives the three input samples they imply and explains the fixture).  The HIP
    tap kernels reproduce the eighteen entries that are predictions to the printed digits, inside a batch as alone."""
    from test_oracle_golden import _t2_flat_kat, t2_kat_frame
    idx, want = _t2_flat_kat()
    x, _samples = t2_kat_frame()
    m = _model("convmodrecnets_CNN2_0.5")
    flat = m.predict(x, tap="flat")
    assert flat.shape == (1, 2580) and np.abs(flat[0, idx] - want).max() < 2.5e-8
    conv = m.predict(x, tap="conv")                                    # model4 = layers[2].output: (1, 2, 129, 10)
    assert np.array_equal(conv.reshape(1, -1), flat)
    xb = np.concatenate([synthetic_frames(37, seed=3, sigma=5e-3), x, synthetic_frames(90, seed=4, sigma=5e-3)])
    assert np.array_equal(m.predict(xb, tap="flat")[37], flat[0])


@pytest.mark.parametrize("name", H5_NAMES)
def test_the_two_unquantised_data_set_frames_the_notebook_prints(name):
    """CNN.ipynb cells 14 / 16 print X_test[2] and X_test[3] in full float precision -- the only unquantised RML2016.10a frames in the
    reference besides cell 18's (tests/golden/notebook_frames.npz).  No Keras output is recorded for them: held to the f64 oracle,
    every checkpoint, every precision mode's label."""
    x = np.load(os.path.join(GOLDEN, "notebook_frames.npz"))["frames"]
    assert x.shape == (2, 2, 128) and 4e-3 < x.std() < 6e-3             # the scale bench.py's synthetic frames are drawn at (sigma 5e-3)
    w = [a for p in load_deployed_npz(name) for a in p]
    ref = O.forward_deployed(x, *w, dtype=np.float64)
    m = _model(name)
    np.testing.assert_allclose(m.predict(x), ref["probs"], atol=2e-6)
    np.testing.assert_allclose(m.predict(x, tap="dense"), ref["dense"], rtol=0, atol=5e-6)
    assert m.predict_classes(x).tolist() == ref["labels"].tolist()
    margin = np.sort(ref["dense"], axis=1)[:, -1] - np.sort(ref["dense"], axis=1)[:, -2]
    for dtype in ("bf16", "f16", "fp8"):
        lab = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"), dtype=dtype).predict_classes(x)
        decided = margin > 0.15 * np.abs(ref["dense"]).max()
        assert (lab[decided] == ref["labels"][decided]).all(), (dtype, lab, ref["labels"], margin)


@pytest.mark.parametrize("name", H5_NAMES)
def test_bundled_frames_all_checkpoints(name):
    x, meta = _frames()
    fz = json.load(open(os.path.join(GOLDEN, "oracle_frozen.json")))["by_weights"][name]
    m = _model(name)
    dense = m.predict(x, tap="dense")
    np.testing.assert_allclose(dense, np.array(fz["dense"]), rtol=0, atol=5e-6)
    assert m.predict_classes(x).tolist() == fz["labels"]
    w = [a for p in load_deployed_npz(name) for a in p]
    ref = O.forward_deployed(x, *w, dtype=np.float64)
    np.testing.assert_allclose(m.predict(x), ref["probs"], atol=2e-6)
    if name.startswith("3conv"):
        i0 = meta["names"].index("12.16.testDataYunyun.txt#0")
        assert np.abs(dense[i0] - np.array([0.0, 3.1391976, 0.3649335])).max() < 5e-3


@pytest.mark.parametrize("name", ["3convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 65536])
def test_synthetic_parity(name, n):
    x = synthetic_frames(n, seed=2016)
    w = [a for p in load_deployed_npz(name) for a in p]
    ref = O.forward_deployed(x, *w, dtype=np.float64)
    m = _model(name)
    scale = max(1.0, float(np.abs(ref["dense"]).max()))
    np.testing.assert_allclose(m.predict(x, tap="dense"), ref["dense"], rtol=0, atol=2e-6 * scale)
    np.testing.assert_allclose(m.predict(x), ref["probs"], rtol=0, atol=2e-6)
    undecided = _check_labels(m.predict_classes(x), ref)
    assert undecided <= max(1, n // 20000)


@pytest.mark.parametrize("filters", [3, 10])
def test_taps_conv_and_flat(filters):
    topo = Topology.deployed(filters)
    w = synthetic_weights(topo, seed=11, bias_scale=0.01)
    x = synthetic_frames(130, seed=5, sigma=0.05)
    ref = O.forward("deployed", x, w, dtype=np.float64)
    m = VTCNN2(topo)
    m.set_weights(w)
    conv = m.predict(x, tap="conv")
    assert conv.shape == (130, 2, 129, filters)
    np.testing.assert_allclose(conv, ref["conv"], rtol=0, atol=1e-6)
    flat = m.predict(x, tap="flat")
    np.testing.assert_array_equal(flat, conv.reshape(130, -1))
    np.testing.assert_allclose(m.predict(x, tap="dense"), ref["dense"], rtol=0, atol=2e-6 * max(1.0, np.abs(ref["dense"]).max()))


def test_batch_size_invariance_and_torch_io():
    m = _model("3convmodrecnets_CNN2_0.5")
    x = synthetic_frames(5000, seed=3, device="cuda")
    a = m.predict(x)
    assert isinstance(a, torch.Tensor) and a.is_cuda and a.shape == (5000, 3)
    for bs in (1, 7, 64, 1024, 4999):
        if bs == 1:
            b = m.predict(x[:300], batch_size=1)
            assert torch.equal(a[:300], b)
        else:
            assert torch.equal(a, m.predict(x, batch_size=bs))
    assert torch.equal(m.predict_classes(x).cpu(), torch.from_numpy(m.predict_classes(x.cpu().numpy())))


def test_first_max_tie_break_and_relu_zero():
    # all-zero input under 5conv: Dense pre-activations are the biases; force exact ties by weights
    topo = Topology.deployed(3)
    w = synthetic_weights(topo, seed=1)
    (ck, cb), (dk, db) = w
    dk[:] = -np.abs(dk)                      # all-negative dense kernel -> z = relu(negative) = 0 exactly
    cb[:] = 0.5
    db[:] = 0.0
    m = VTCNN2(topo)
    m.set_weights([(ck, cb), (dk, db)])
    x = synthetic_frames(257, seed=9)
    assert (m.predict(x, tap="dense") == 0).all()
    assert (m.predict_classes(x) == 0).all()
    np.testing.assert_array_equal(m.predict(x), np.full((257, 3), np.float32(1 / 3)))
    # exact two-way tie between classes 1 and 2 (identical columns), class 0 lower -> label 1
    dk[:, 2] = dk[:, 1]
    db[:] = [0.0, 100.0, 100.0]
    m.set_weights([(ck, cb), (dk, db)])
    d = m.predict(x, tap="dense")
    assert (d[:, 1] == d[:, 2]).all() and (d[:, 1] > d[:, 0]).all()
    assert (m.predict_classes(x) == 1).all()


def test_empty_and_errors():
    m = _model("3convmodrecnets_CNN2_0.5")
    assert m.predict(np.zeros((0, 2, 128), np.float32)).shape == (0, 3)
    with pytest.raises(ValueError):
        m.predict(np.zeros((4, 2, 127), np.float32))
    with pytest.raises(ValueError):
        m.predict(np.zeros((1, 2, 128), np.float32), tap="nonsense")
    with pytest.raises(ValueError):
        VTCNN2(Topology.deployed(3)).set_weights(load_deployed_npz("convmodrecnets_CNN2_0.5"))
    from modulationdetectioncnn_amd._cabi import MdcError
    with pytest.raises(MdcError):       # fp8: vtcnn2 and deployed, not the cnn.py literal model
        VTCNN2.synthetic("cnnpy", classes=5, dtype="fp8").predict(np.zeros((1, 2, 128), np.float32))
    with pytest.raises(MdcError):       # bf16: vtcnn2 and deployed, not the cnn.py literal model
        VTCNN2.synthetic("cnnpy", classes=5, dtype="bf16").predict(np.zeros((1, 2, 128), np.float32))
    with pytest.raises(MdcError):       # f16: the deployed nets only
        VTCNN2.synthetic("vtcnn2", classes=11, dtype="f16").predict(np.zeros((1, 2, 128), np.float32))


def test_txt_weights_load_and_classify(tmp_path):
    """Q6.12 text export -> same labels as the .h5 weights on the bundled frames (weights differ by < 1 LSB)."""
    from modulationdetectioncnn_amd.formats import q612
    z = np.load(os.path.join(GOLDEN, "weights_txt", "12.15.latestWeights.npz"))
    m = VTCNN2(Topology.deployed(3))
    m.set_weights([(z["conv_kernel"], z["conv_bias"]), (z["dense_kernel"], z["dense_bias"])])
    x, _ = _frames()
    fz = json.load(open(os.path.join(GOLDEN, "oracle_frozen.json")))["by_weights"]["3convmodrecnets_CNN2_0.5"]
    assert m.predict_classes(x).tolist() == fz["labels"]
    w = q612.DeployedWeights(3, z["conv_kernel"], z["conv_bias"], z["dense_kernel"], z["dense_bias"])
    p = tmp_path / "w.txt"
    p.write_text(q612.dump_weights_f3(w))
    m2 = VTCNN2.from_txt(str(p))
    assert m2.predict_classes(x).tolist() == fz["labels"]


def test_confusion_and_accuracy_match_reference_loop():
    """cnn.py:199-216,257-259 on top of predict(): confusion matrix and accuracy."""
    m = _model("3convmodrecnets_CNN2_0.5")
    x = synthetic_frames(999, seed=21)
    y = np.arange(999) % 3
    w = [a for p in load_deployed_npz("3convmodrecnets_CNN2_0.5") for a in p]
    ref = O.forward_deployed(x, *w, dtype=np.float64)
    np.testing.assert_allclose(m.confusion(x, y), O.confusion(y, ref["labels"], 3), atol=1e-12)
    assert abs(m.accuracy(x, y) - float((ref["labels"] == y).mean())) < 1e-12


# ---------------------------------------------------------------------------------------------------------------
# bf16 mode (csrc/deployed_bf16.hip): the dense layer on the matrix cores, conv outputs and dense weights rounded to
# bf16.  Parity bar (stated here, checked below): class sums within 1e-2 of the largest |class sum| of the frame
# (observed: a few 1e-3), probabilities within 1e-2 abs, labels equal wherever the f64 oracle's top-2 margin
# exceeds 2e-2 of the largest class sum.  T2 has no recorded reference outputs (parity unpinned beyond the oracle).
BF16_TOL = 1e-2


def _bf16_check(name, x, dtype="bf16"):
    w = [a for p in load_deployed_npz(name) for a in p]
    ref = O.forward_deployed(x, *w, dtype=np.float64)
    m = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"), dtype=dtype)
    probs = m.predict(x)
    lab = m.predict_classes(x)
    np.testing.assert_allclose(probs, ref["probs"], atol=BF16_TOL)
    assert np.isfinite(probs).all() and np.abs(probs.sum(axis=1) - 1).max() < 1e-5
    d = ref["dense"]
    srt = np.sort(d, axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 2 * BF16_TOL * np.maximum(np.abs(d).max(axis=1), 1e-30)
    assert (lab[decided] == ref["labels"][decided]).all()
    return m, probs, lab, float(decided.mean())


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("name", ["3convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5"])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 127, 1000, 65536])
def test_bf16_mode_parity(name, n, dtype):
    """Both 16-bit modes against the same bar (f16 -- 11 significant bits, conv in packed f16 -- lands well inside it)."""
    x = synthetic_frames(n, seed=2016)
    _bf16_check(name, x, dtype)


def test_f16_mode_is_tighter_than_bf16():
    """f16 operands keep three more bits than bf16: its class sums must sit closer to the f64 oracle."""
    name = "convmodrecnets_CNN2_0.5"
    x = synthetic_frames(4096, seed=31)
    w = [a for p in load_deployed_npz(name) for a in p]
    ref = O.forward_deployed(x, *w, dtype=np.float64)["probs"]
    err = {}
    for dt in ("bf16", "f16"):
        m = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"), dtype=dt)
        err[dt] = float(np.abs(m.predict(x) - ref).max())
    assert err["f16"] < err["bf16"] and err["f16"] < 3e-3, err


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("name", H5_NAMES)
def test_bf16_mode_bundled_frames_keep_their_frozen_labels(name, dtype):
    """The reference's own frames through the bf16 kernels: same labels as the frozen f64-oracle labels wherever the
    decision is not a near-tie, class sums within the bar."""
    x, _meta = _frames()
    _bf16_check(name, x, dtype)


def test_bf16_mode_scale_bias_and_edges():
    """Large inputs (x400, the bf16 range is f32's), a frame of zeros, and an impulse at each row end / row start:
    exercises the zero padding at w = 0 and w = 128 and the piece boundaries of the lane = frame layout."""
    name = "convmodrecnets_CNN2_0.5"
    x = synthetic_frames(64, seed=5) * 400.0
    x[1] = 0.0
    for i, (h, s) in enumerate([(0, 0), (0, 127), (1, 0), (1, 127), (0, 3), (0, 4), (0, 63), (0, 64), (1, 31), (1, 32)]):
        x[2 + i] = 0.0
        x[2 + i, h, s] = 1.0
    for dt in ("bf16", "f16"):          # x400 keeps the conv outputs (~1e2) far below the f16 range
        _bf16_check(name, x, dt)
        _bf16_check("3convmodrecnets_CNN2_0.5", x, dt)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_bf16_mode_is_independent_of_batch_composition_and_rejects_taps(dtype):
    from modulationdetectioncnn_amd import _cabi
    name = "convmodrecnets_CNN2_0.5"
    m = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"), dtype=dtype)
    x = synthetic_frames(1000, seed=8)
    whole = m.predict(x)
    np.testing.assert_array_equal(m.predict(x, batch_size=37), whole)                 # ragged chunks
    np.testing.assert_array_equal(m.predict(x[5:22]), whole[5:22])                    # other neighbours in the 16-frame group
    perm = np.random.default_rng(0).permutation(1000)
    np.testing.assert_array_equal(m.predict(x[perm]), whole[perm])
    with pytest.raises(_cabi.MdcError):
        m.predict(x[:4], tap="dense")
    # raw bytes: the 16-bit kernels read them themselves; same results as converting first
    iq = np.random.default_rng(1).integers(0, 256, size=256 * 33, dtype=np.uint8)
    from modulationdetectioncnn_amd import frames_from_iq_u8
    p, _l = m.predict_iq_u8(iq, 0.02 / 127.5)
    np.testing.assert_array_equal(p, m.predict(frames_from_iq_u8(iq, 0.02 / 127.5)).cpu().numpy())


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_16bit_modes_keep_an_out_of_range_frame_to_itself(dtype):
    """A frame far outside the f16 range (x = 1e6: conv outputs overflow f16 to inf) or full of NaN must not change
    any other frame of its 16-frame group: MFMA columns are independent, zero weights never meet the bad values of
    another frame."""
    name = "convmodrecnets_CNN2_0.5"
    m = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"), dtype=dtype)
    x = synthetic_frames(64, seed=13)
    clean = m.predict(x)
    bad = x.copy()
    bad[5] *= 2e8
    bad[20] = np.nan
    out = m.predict(bad)
    keep = np.ones(64, bool)
    keep[[5, 20]] = False
    np.testing.assert_array_equal(out[keep], clean[keep])
    if dtype == "bf16":          # bf16 has f32's range: a huge frame still classifies (softmax saturates to one class)
        assert np.isfinite(out[5]).all() and abs(out[5].sum() - 1) < 1e-5


# ---------------------------------------------------------------------------------------------------------------
# The f32 variant with Dense(3) on the f32 matrix pipe (csrc/deployed_f32m.hip): same bar as the production f32 kernel.
# It is kept as the measured answer to "can the dense layer leave the VALU at f32?" (HISTORY.md section 4.1c): correct to
# the same tolerances, slower -- so it lives in the alternates test build only (libmdc_alt.so, -DMDC_ALTERNATES), where
# mdc_create reads MDC_DEP_F32_MFMA=1 once per model.
@pytest.fixture
def f32_mfma(monkeypatch):
    import sys
    monkeypatch.setenv("MDC_DEP_F32_MFMA", "1")       # read by libmdc_alt.so when a model is created
    monkeypatch.setattr(sys.modules[__name__], "_VARIANT", "alternates")


@pytest.mark.parametrize("ring", [2, 3, 4, 6, 8])
def test_lds_dma_ring_form_of_the_t1_kernel_is_bit_identical(monkeypatch, ring):
    """Round 3: T1's f32 kernel fed through a per-wave LDS-DMA ring (asm-issued copies, counted vmcnt)
    instead of direct loads -- measured slower at every depth (HISTORY.md 4.1), so it lives in the alternates build; its
    results must be the product's bits: full blocks through the ring, the ragged tail through the tail form, more blocks
    than waves and fewer."""
    monkeypatch.setenv("MDC_DEP_RING", str(ring))
    name = "3convmodrecnets_CNN2_0.5"
    ref = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"))
    alt = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"), _lib_variant="alternates")
    for n in (64, 65, 1000, 64 * 2049 + 7, 1 << 19):
        x = synthetic_frames(n, seed=n, device="cuda")
        pa, la, _ = alt.forward_device(x)
        pr, lr, _ = ref.forward_device(x)
        assert torch.equal(pa, pr) and torch.equal(la, lr), (ring, n)


def test_the_alternates_build_runs_the_product_kernels_unless_told_otherwise():
    """libmdc_alt.so without any of its environment variables is the product: bit-identical results."""
    x = synthetic_frames(1000, seed=4)
    for name in ("3convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5"):
        a = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz")).predict(x)
        b = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"), _lib_variant="alternates").predict(x)
        np.testing.assert_array_equal(a, b)


def test_f32_mfma_variant_keras_known_answer_and_bundled_frames(f32_mfma):
    k = json.load(open(os.path.join(GOLDEN, "keras_kat.json")))
    x = np.asarray(k["input"], np.float32).reshape(1, 2, 128)
    m = _model("3convmodrecnets_CNN2_0.5")
    assert np.abs(m.predict(x, tap="dense")[0] - np.array(k["keras_dense"])).max() < 5e-6
    assert m.predict_classes(x).tolist() == [0]
    xb, _ = _frames()
    for name in H5_NAMES:
        fz = json.load(open(os.path.join(GOLDEN, "oracle_frozen.json")))["by_weights"][name]
        mm = _model(name)
        np.testing.assert_allclose(mm.predict(xb, tap="dense"), np.array(fz["dense"]), rtol=0, atol=5e-6)
        assert mm.predict_classes(xb).tolist() == fz["labels"]


@pytest.mark.parametrize("name", ["3convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5"])
@pytest.mark.parametrize("n", [1, 3, 4, 5, 63, 64, 65, 1000, 70001])
def test_f32_mfma_variant_synthetic_parity_and_invariances(f32_mfma, name, n):
    x = synthetic_frames(n, seed=2016)
    w = [a for p in load_deployed_npz(name) for a in p]
    ref = O.forward_deployed(x, *w, dtype=np.float64)
    m = _model(name)
    scale = max(1.0, float(np.abs(ref["dense"]).max()))
    np.testing.assert_allclose(m.predict(x, tap="dense"), ref["dense"], rtol=0, atol=2e-6 * scale)
    p = m.predict(x)
    np.testing.assert_allclose(p, ref["probs"], rtol=0, atol=2e-6)
    lab = m.predict_classes(x)
    assert _check_labels(lab, ref) <= max(1, n // 20000)
    assert (lab == np.argmax(p, axis=1)).all()
    if n >= 63:
        # chunking and permutation leave every bit alone (runs of 1..16 four-frame groups, any position in a group)
        xt = torch.from_numpy(x).cuda()
        pt = m.predict(xt)
        for bs in (1, 5, 64, 997):
            if bs == 1 and n > 2000:
                continue
            assert torch.equal(pt, m.predict(xt, batch_size=bs))
        perm = torch.randperm(n, device="cuda")
        assert torch.equal(m.predict(xt[perm].contiguous()), pt[perm])


def test_f32_mfma_variant_raw_bytes_and_nonfinite_isolation(f32_mfma):
    from modulationdetectioncnn_amd import frames_from_iq_u8
    for name in ("3convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5"):
        m = _model(name)
        for hop, n in ((128, 4097), (3, 1000), (1, 65)):
            iq = torch.from_numpy(np.random.default_rng(hop).integers(0, 256, size=2 * (128 + hop * (n - 1)), dtype=np.uint8)).cuda()
            p, l = m.predict_iq_u8(iq, 0.02 / 127.5, hop=hop)
            p2, l2, _ = m.forward_device(frames_from_iq_u8(iq, 0.02 / 127.5, hop=hop))
            assert torch.equal(p, p2) and torch.equal(l, l2), (name, hop)
        x = synthetic_frames(4096, seed=5, device="cuda")
        base = m.predict(x)
        xb = x.clone()
        bad = torch.tensor([1, 2, 7, 64, 4000], device="cuda")
        xb[bad, 0, 5] = float("inf")
        xb[bad[::2], 1, 127] = float("nan")
        keep = torch.ones(4096, dtype=torch.bool, device="cuda")
        keep[bad] = False
        assert torch.equal(m.predict(xb)[keep], base[keep])


# ---------------------------------------------------------------------------------------------------------------
# fp8 mode of the deployed nets (round 2; BASELINE configs[4] read literally: "5convmodrecnets_CNN2_0.5.wts.h5, fp8 MFMA
# path").  e4m3 keeps 4 significant bits: the bar is stated here -- probabilities within 6e-2 of the f64 oracle, labels
# equal to the f32 kernel's on >= 97 % of 65,536 synthetic frames -- and it is parity-unpinned like every reduced mode.
@pytest.mark.parametrize("name", ["5convmodrecnets_CNN2_0.5", "3convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5"])
def test_fp8_mode_of_the_deployed_nets(name):
    from modulationdetectioncnn_amd import frames_from_iq_u8
    w = load_deployed_npz(name)
    flat = [a for p in w for a in p]
    m8 = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"), dtype="fp8")
    mf = _model(name)
    for n in (1, 15, 16, 17, 1000):
        x = synthetic_frames(n, seed=100 + n)
        ref = O.forward_deployed(x, *flat, dtype=np.float64)
        p = m8.predict(x)
        assert p.shape == (n, 3) and np.abs(p - ref["probs"]).max() < 6e-2, (name, n, np.abs(p - ref["probs"]).max())
        np.testing.assert_allclose(p.sum(axis=1), 1.0, atol=1e-5)
        assert (m8.predict_classes(x) == np.argmax(p, axis=1)).all()
    xg = synthetic_frames(1 << 16, seed=77, device="cuda")
    agree = float((m8.predict_classes(xg) == mf.predict_classes(xg)).float().mean())
    assert agree >= 0.97, (name, agree)
    # frames independent of their neighbours, chunking, permutation: bit for bit
    whole = m8.predict(xg[:5000])
    assert torch.equal(whole, m8.predict(xg[:5000], batch_size=37))
    perm = torch.randperm(5000, device="cuda")
    assert torch.equal(m8.predict(xg[:5000][perm].contiguous()), whole[perm])
    # the reference's bundled frames (x up to 0.02: the default fp8_input_absmax)
    xb, _ = _frames()
    refb = O.forward_deployed(xb, *flat, dtype=np.float64)
    assert np.abs(m8.predict(xb) - refb["probs"]).max() < 6e-2
    # raw bytes at a hop, same mode: bit-identical to convert-then-forward
    iq = torch.from_numpy(np.random.default_rng(2).integers(0, 256, size=2 * (128 + 7 * 999), dtype=np.uint8)).cuda()
    p1, l1 = m8.predict_iq_u8(iq, 0.02 / 127.5, hop=7)
    p2, l2, _ = m8.forward_device(frames_from_iq_u8(iq, 0.02 / 127.5, hop=7))
    assert torch.equal(p1, p2) and torch.equal(l1, l2)
    # beyond the stated range the activations saturate at the e4m3 maximum: finite, rows sum to 1, no leak into neighbours
    xs = synthetic_frames(64, seed=3)
    clean = m8.predict(xs)
    xs2 = xs.copy()
    hot = [1.6, 2.0, 2.2, 2.4, 3.0, 4.0, 8.0, 1e4]      # the scale leaves a factor 2 of head-room: 2.0-2.3 lands conv outputs in (448, 512),
    for i, sc in enumerate(hot):                      # where only MODE.FP16_OVFL keeps the conversion from returning NaN (deployed_bf16.hip)
        xs2[5 + i] = synthetic_frames(1, seed=40 + i, sigma=0.02)[0].clip(-0.02, 0.02) * np.float32(sc)
    out = m8.predict(xs2)
    assert np.isfinite(out).all() and np.abs(out[5:5 + len(hot)].sum(axis=1) - 1).max() < 1e-5
    keep = np.ones(64, bool)
    keep[5:5 + len(hot)] = False
    np.testing.assert_array_equal(out[keep], clean[keep])
    with pytest.raises(Exception):
        m8.predict(xs[:4], tap="dense")
    # a model told its range classifies larger inputs
    mbig = VTCNN2.from_npz(os.path.join(GOLDEN, "weights", name + ".npz"), dtype="fp8", fp8_input_absmax=2.0)
    xl = synthetic_frames(512, seed=9, sigma=0.5)
    refl = O.forward_deployed(xl, *flat, dtype=np.float64)
    assert np.abs(mbig.predict(xl) - refl["probs"]).max() < 8e-2


# ---- f32: the ReLU rides in the clamp bit of the conv's second fma, on a table scaled by exact powers of two (deployed.hip)
@pytest.mark.parametrize("name", ["3convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5"])
@pytest.mark.parametrize("n", [1, 63, 65, 70001])
def test_clamp_relu_conv_meets_the_oracle_and_the_unscaled_tap_kernel(name, n):
    """The fast kernel's table carries taps/bias x 2^-32 and dense weights x 2^+32; the one-frame-at-a-time tap kernel
    reads the UNSCALED table and takes fmaxf.  Both are held to the f64 oracle at the f32 bar, and the conv activations the
    tap kernel reports are the ones the fast kernel's probabilities are consistent with."""
    x = synthetic_frames(n, seed=77) * np.float32(2.5)
    w = [a for p in load_deployed_npz(name) for a in p]
    ref = O.forward_deployed(x, *w, dtype=np.float64)
    scale = max(1.0, float(np.abs(ref["dense"]).max()))
    m = _model(name)
    d, p, l = m.predict(x, tap="dense"), m.predict(x), m.predict_classes(x)
    np.testing.assert_allclose(d, ref["dense"], rtol=0, atol=2e-6 * scale)
    np.testing.assert_allclose(p, ref["probs"], rtol=0, atol=2e-6)
    _check_labels(l, ref)
    conv = m.predict(x[:64], tap="conv")      # unscaled table, plain max
    np.testing.assert_allclose(conv, ref["conv"][:64], rtol=0, atol=2e-6 * max(1.0, float(np.abs(ref["conv"]).max())))


def test_clamp_relu_conv_with_zero_tiny_and_huge_taps():
    """Zero and denormal-after-scaling taps and conv biases far above the activations: the scaled table must stay the same
    function (documented range of exactness: |tap|, |bias| >= 2^-94 or 0, activations < 2^32)."""
    topo = Topology.deployed(10, 3)
    (ck, cb), (dk, db) = synthetic_weights(topo, seed=5, bias_scale=0.05)
    ck = np.array(ck, np.float32)
    ck.reshape(2, 10)[1, 3] = 0.0
    ck.reshape(2, 10)[1, 7] = 1e-30          # x 2^-32 is an f32 denormal: its contribution (~1e-30) is below every bar
    x = synthetic_frames(5000, seed=8) * np.float32(3.0)
    ref = O.forward_deployed(x, ck, cb, dk, db, dtype=np.float64)
    m = VTCNN2(topo)
    m.set_weights([(ck, cb), (dk, db)])
    np.testing.assert_allclose(m.predict(x), ref["probs"], rtol=0, atol=2e-6)
    (ck2, cb2), (dk2, db2) = synthetic_weights(topo, seed=6, bias_scale=0.05)
    cb2 = np.full(10, 5.0, np.float32)
    ref2 = O.forward_deployed(x, ck2, cb2, dk2, db2, dtype=np.float64)
    m = VTCNN2(topo)
    m.set_weights([(ck2, cb2), (dk2, db2)])
    d = m.predict(x, tap="dense")
    np.testing.assert_allclose(d, ref2["dense"], rtol=0, atol=2e-6 * max(1.0, float(np.abs(ref2["dense"]).max())))
    # activations of 1e6 (inputs x 1e6): far inside the clamp's 2^32 ceiling
    xb = x[:512] * np.float32(1e5)
    refb = O.forward_deployed(xb, ck, cb, dk, db, dtype=np.float64)
    m2 = VTCNN2(topo)
    m2.set_weights([(ck, cb), (dk, db)])
    got = m2.predict(xb, tap="dense")
    np.testing.assert_allclose(got, refb["dense"], rtol=0, atol=4e-6 * max(1.0, float(np.abs(refb["dense"]).max())))


def test_activations_beyond_the_documented_range_saturate_at_2_pow_32():
    """include/mdc.h: conv activations of 2^32 and beyond saturate (the clamp bit's [0, 1] on the 2^-32-scaled table).  Pinned
    here so the behaviour cannot drift: frames 1e12 times the bundled level give the f64 oracle's result with its conv
    output clipped at 2^32, finite probabilities, and leave the other frames of the batch untouched."""
    name = "3convmodrecnets_CNN2_0.5"
    w = [a for p in load_deployed_npz(name) for a in p]
    x = synthetic_frames(256, seed=21)
    big = x.copy()
    big[100:104] *= np.float32(1e12)
    ref = O.forward_deployed(big.astype(np.float64), *w, dtype=np.float64)
    flat = np.minimum(ref["conv"], 2.0 ** 32).reshape(256, -1)
    dense = np.maximum(flat @ np.asarray(w[2], np.float64) + np.asarray(w[3], np.float64), 0.0)
    m = _model(name)
    got = m.predict(big, tap="dense")
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got[100:104], dense[100:104], rtol=2e-6)
    clean = m.predict(x, tap="dense")
    keep = np.ones(256, bool)
    keep[100:104] = False
    np.testing.assert_array_equal(got[keep], clean[keep])
    p = m.predict(big)
    assert np.isfinite(p).all() and np.abs(p.sum(axis=1) - 1).max() < 1e-5


def test_activations_below_the_documented_range_lose_bits_and_taps_say_so():
    """ADVICE r3 / include/mdc.h: bit-identity to Keras' operation order holds for conv activations in [2^-94, 2^32).  The
    lower end, pinned: with zero biases and frames around 2^-100 the product kernel's 2^-32-scaled activations are f32
    denormals (or flushed) -- its dense output is within 2^-94 * sum|w| of the truth, not within f32 relative precision --
    while the MDC_TAP_CONV kernel (unscaled table, fmaxf) still gives the activations to f32 precision: tap and
    probabilities may disagree outside the range, as the header says.  An Inf sample yields FINITE probabilities (Keras: NaN)
    and touches no other frame."""
    topo = Topology.deployed(3, 3)
    rng = np.random.default_rng(5)
    ck = rng.normal(0, 0.5, (1, 2, 1, 3)).astype(np.float32)
    cb = np.zeros(3, np.float32)
    dk = rng.normal(0, 0.05, (774, 3)).astype(np.float32)
    db = np.zeros(3, np.float32)
    m = VTCNN2(topo)
    m.set_weights([(ck, cb), (dk, db)])
    x = (synthetic_frames(64, seed=4, sigma=1.0) * np.float32(2.0 ** -100)).astype(np.float32)
    ref = O.forward_deployed(x.astype(np.float64), ck, cb, dk, db, dtype=np.float64)
    conv = m.predict(x, tap="conv")
    np.testing.assert_allclose(conv, ref["conv"], rtol=0, atol=2e-6 * float(np.abs(ref["conv"]).max()))      # the tap kernel: f32 rounding at THIS scale
    dense = m.predict(x, tap="dense")
    assert np.isfinite(dense).all() and (dense >= 0).all()
    bound = 2.0 ** -94 * float(np.abs(dk).sum(axis=0).max())
    assert float(np.abs(dense - ref["dense"]).max()) <= bound                          # absolute, from the denormal grid
    assert float(np.abs(ref["dense"]).max()) < 64 * bound                              # ... i.e. no relative precision is claimed here
    # inside the range the same model is bit-level exact again
    x2 = (x * np.float32(2.0 ** 40)).astype(np.float32)
    ref2 = O.forward_deployed(x2.astype(np.float64), ck, cb, dk, db, dtype=np.float64)
    np.testing.assert_allclose(m.predict(x2, tap="dense"), ref2["dense"], rtol=0, atol=4e-6 * float(np.abs(ref2["dense"]).max()))
    # Inf: finite probabilities from the product kernel, neighbours untouched
    xi = synthetic_frames(64, seed=4)
    clean = m.predict(xi)
    xi[7, 0, 5] = np.inf
    p = m.predict(xi)
    assert np.isfinite(p).all()
    keep = np.arange(64) != 7
    np.testing.assert_array_equal(p[keep], clean[keep])


def test_conv_taps_of_both_signs_with_biases_and_saturating_inputs():
    """both signs of the second tap, biases of both signs and a saturating input scale"""
    topo = Topology.deployed(10, 3)
    rng = np.random.default_rng(11)
    ck = rng.normal(0, 1, (1, 2, 1, 10)).astype(np.float32)
    ck.reshape(2, 10)[1] = np.array([-2, -1, -0.5, -0.25, -3, 2, 1, 0.5, 0.25, 3], np.float32)
    cb = rng.normal(0, 0.05, 10).astype(np.float32)
    dk = rng.normal(0, 0.05, (2580, 3)).astype(np.float32)
    db = np.array([0.3, -0.2, 0.1], np.float32)
    m = VTCNN2(topo)
    m.set_weights([(ck, cb), (dk, db)])
    for s in (0.01, 1.0, 30.0):
        x = synthetic_frames(4096, seed=3, sigma=1.0) * np.float32(s)
        ref = O.forward_deployed(x, ck, cb, dk, db, dtype=np.float64)
        scale = max(1.0, float(np.abs(ref["dense"]).max()))
        np.testing.assert_allclose(m.predict(x, tap="dense"), ref["dense"], rtol=0, atol=4e-6 * scale)
        _check_labels(m.predict_classes(x), ref)
