"""The training step on the MI355X (csrc/train.hip via mdc_trainer_*) against oracle/oracle_train.py.

PARITY UNPINNED: the reference bundles no dataset and records no training run; the GPU is held to
this repo's numpy oracle, which tests/test_oracle_train.py holds to torch autograd and to TensorFlow's published Adam.
Bars: gradients of a 1,024-frame batch <= 1e-5 of the largest entry of each tensor (GPU f32 against the f64 oracle);
a 20-step Adam trajectory <= 1e-4 absolute on the weights (they are O(0.1)); losses <= 1e-5 relative."""
import json
import os

import numpy as np
import pytest

from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_weights
from modulationdetectioncnn_amd import _cabi
from modulationdetectioncnn_amd.formats import q612
from modulationdetectioncnn_amd.formats.h5mini import H5File, load_keras_h5
from modulationdetectioncnn_amd.training import Trainer, to_onehot
from oracle import oracle_np as O
from oracle import oracle_train as T
from tests.signals import modulated_frames

pytestmark = pytest.mark.gpu

CASES = [("deployed", Topology.deployed(3, 3)), ("deployed", Topology.deployed(10, 3)), ("cnnpy", Topology.cnnpy(10, 10, 5)),
         ("cnnpy", Topology.cnnpy(4, 7, 3)), ("cnnpy", Topology.cnnpy(10, 16, 16))]
IDS = ["T1-F3", "T2-F10", "T4-cnnpy", "T4-4-7-3", "T4-10-16-16"]


def _data(topo, n, seed, gain=None):
    x, lab, _ = modulated_frames(n, seed=seed)
    if gain is None:
        gain = 40.0 if topo.kind == "cnnpy" else 1.0      # T4's random-init logits are ~0 at the frames' native 1e-2 scale
    x = (x * gain).astype(np.float32)
    return x, to_onehot(lab % topo.classes, topo.classes)


def _leveled(n, seed):
    """Signal-shaped frames whose three classes also differ in LEVEL (x0.4 / x1 / x2.2).  The deployed net is a rectified
    two-tap filter with a linear read-out: what it can learn from raw I/Q is envelope statistics, and tests/signals.py
    normalises every frame to one rms level (RadioML's AM-SSB / WBFM / GFSK bursts do differ in envelope; the dataset is
    not here).  This is a separable stand-in, not a claim about accuracy on RML2016.10a."""
    x, lab, _ = modulated_frames(n, seed=seed)
    return (x * np.array([0.4, 1.0, 2.2], np.float32)[lab][:, None, None]).astype(np.float32), lab


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("kind,topo", CASES, ids=IDS)
def test_gradients_of_a_1024_frame_batch(kind, topo):
    w = synthetic_weights(topo, seed=3, bias_scale=0.05)
    x, y = _data(topo, 1024, seed=17)
    tr = Trainer(topo, w, device=0)
    loss, grads = tr.loss_and_gradients(x, y)
    ref_loss, _li, ref_grads, _p = T.loss_and_grads(kind, x, y, w, np.float64)
    assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss)
    for (gk, gb), (rk, rb) in zip(grads, ref_grads):
        assert np.abs(rk).max() > 0
        assert _rel(gk, rk) <= 1e-5 and _rel(gb, rb) <= 1e-5, (_rel(gk, rk), _rel(gb, rb))
    # nothing was applied
    for (k, b), (k0, b0) in zip(tr.get_weights(), w):
        assert np.array_equal(k, k0) and np.array_equal(b, b0)
    assert tr.read()["iterations"] == 0


@pytest.mark.parametrize("kind,topo", CASES[:3], ids=IDS[:3])
@pytest.mark.parametrize("n", [1, 3, 63, 257, 5000])
def test_ragged_batches(kind, topo, n):
    w = synthetic_weights(topo, seed=4, bias_scale=0.05)
    x, y = _data(topo, n, seed=n)
    loss, grads = Trainer(topo, w, device=0).loss_and_gradients(x, y)
    ref_loss, _li, ref_grads, _p = T.loss_and_grads(kind, x, y, w, np.float64)
    assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss)
    for (gk, gb), (rk, rb) in zip(grads, ref_grads):
        assert _rel(gk, rk) <= 1e-5 and _rel(gb, rb) <= 2e-5


@pytest.mark.parametrize("kind,topo", CASES[:3], ids=IDS[:3])
def test_adam_trajectory_of_20_steps(kind, topo):
    w = synthetic_weights(topo, seed=6, bias_scale=0.02)
    x, y = _data(topo, 1024, seed=23)
    tr = Trainer(topo, w, device=0)
    xd, yd = tr._frames(x), tr._targets(y, len(x))
    wo = [(k.copy(), b.copy()) for k, b in w]
    opt = T.KerasAdam([t.shape for t in T.flatten_weights(wo)])
    ref_losses = []
    for _ in range(20):
        tr.train_batch(xd, yd, apply=True)
        ref_losses.append(T.train_step(kind, x, y, wo, opt, np.float64))
    r = tr.read()
    assert r["iterations"] == 20 and r["train_frames"] == 20 * 1024
    assert abs(r["train_loss_sum"] / 1024 - sum(ref_losses)) <= 1e-5 * sum(ref_losses)
    assert ref_losses[-1] < ref_losses[0]
    for (k, b), (rk, rb), (k0, _b0) in zip(tr.get_weights(), wo, w):
        assert np.abs(k - rk).max() <= 1e-4 and np.abs(b - rb).max() <= 1e-4
        assert np.abs(k - k0).max() > 5e-3      # 20 steps of ~lr each: the weights moved 100x further than the bar
    st = tr.optimizer_state()
    # the moments are running means of gradients taken along two trajectories (f32 here, f64 there) that drift apart by up
    # to 1e-4 in the weights: a loose bar (measured 5e-3 on the 10-filter net's conv bias), there to catch a wrong formula
    for (mk, mb), (rm_k, rm_b) in zip(st["m"], zip(opt.m[0::2], opt.m[1::2])):
        assert _rel(mk, rm_k) <= 2e-2 and _rel(mb, rm_b) <= 2e-2
    for (vk, vb), (rv_k, rv_b) in zip(st["v"], zip(opt.v[0::2], opt.v[1::2])):
        assert _rel(vk, rv_k) <= 2e-2 and _rel(vb, rv_b) <= 2e-2


def test_a_step_is_reproducible_bit_for_bit_and_the_shuffle_is_an_index_array():
    import torch
    topo = Topology.deployed(3)
    w = synthetic_weights(topo, seed=1)
    x, y = _data(topo, 3000, seed=2)
    perm = np.random.default_rng(0).permutation(3000).astype(np.int32)
    outs = []
    for variant in range(3):
        tr = Trainer(topo, w, device=0)
        if variant < 2:      # order array over the resident set
            xd, yd, od = tr._frames(x), tr._targets(y, 3000), torch.from_numpy(perm).cuda()
            for s in range(0, 3000, 1024):
                tr.train_batch(xd, yd, od, s, min(1024, 3000 - s))
        else:                # the same batches gathered on the host
            for s in range(0, 3000, 1024):
                idx = perm[s:s + 1024]
                tr.train_batch(tr._frames(x[idx]), tr._targets(y[idx], len(idx)))
        outs.append((tr.get_weights(), tr.read()))
    for (wa, ra), (wb, rb) in zip(outs, outs[1:]):
        assert ra == rb
        for (k, b), (k2, b2) in zip(wa, wb):
            assert np.array_equal(k, k2) and np.array_equal(b, b2)
    assert outs[0][1]["iterations"] == 3 and outs[0][1]["train_frames"] == 3000


@pytest.mark.parametrize("kind,topo", [CASES[0], CASES[2]], ids=[IDS[0], IDS[2]])
def test_evaluate_is_the_inference_paths_evaluate(kind, topo):
    w = synthetic_weights(topo, seed=9, bias_scale=0.05)
    x, y = _data(topo, 4097, seed=31)
    tr = Trainer(topo, w, device=0)
    v = tr.evaluate(x, y)
    assert abs(v - T.evaluate(kind, x, y, w, np.float64)) <= 1e-5 * v
    m = VTCNN2(topo, device=0)
    m.set_weights(w)
    assert abs(v - m.evaluate(x, y)) <= 1e-5 * v      # mdc_forward + mdc_crossentropy


def test_fit_learns_three_modulations_and_keeps_the_callbacks_semantics(tmp_path):
    """(c) of the round's bar: on signal-shaped 3-class frames the loss falls and validation accuracy clears chance; the
    best checkpoint is the arg-min of val_loss (strict <), patience counts epochs without improvement."""
    topo = Topology.deployed(3)
    x, lab = _leveled(6000, seed=77)
    xt, yt, xv, yv = x[:4200], lab[:4200], x[4200:], lab[4200:]         # CNN.ipynb cell 4: a 70/30 split
    m = VTCNN2.synthetic(topo, seed=12, device=0)
    m.compile(loss="categorical_crossentropy", optimizer="adam", lr=0.01)      # (10x Keras' default: 12 epochs must suffice here)
    acc0 = m.accuracy(xv, yv)
    ck = str(tmp_path / "best.wts.h5")
    h = m.fit(xt, to_onehot(yt, 3), batch_size=1024, epochs=12, validation_data=(xv, to_onehot(yv, 3)), patience=5,
              checkpoint=ck, seed=5)
    loss, val = h.history["loss"], h.history["val_loss"]
    assert h.epoch == list(range(len(loss))) and len(val) == len(loss)
    assert loss[-1] < 0.8 * loss[0] and min(val) < val[0]
    assert h.best_epoch == int(np.argmin(val))
    if h.stopped_epoch is not None:
        assert h.stopped_epoch == h.best_epoch + 5
    # Keras leaves the LAST weights in the model; the file holds the best (cnn.py:147 loads them back)
    last = m.get_weights()
    assert abs(m.evaluate(xv, yv) - val[-1]) <= 1e-5 * val[-1]
    m.load_weights(ck)
    assert abs(m.evaluate(xv, yv) - min(val)) <= 1e-5 * min(val)
    for (k, b), (k2, b2) in zip(m.get_weights(), h.best_weights):
        assert np.array_equal(k, k2) and np.array_equal(b, b2)
    assert m.accuracy(xv, yv) > max(0.45, acc0)            # chance is 1/3
    # the checkpoint is a full-model save: Adam's state at the best epoch
    f = H5File(ck)
    it = int(f.read("optimizer_weights/Adam/iter:0"))
    assert it == (h.best_epoch + 1) * 5                    # 4,200 frames / 1,024 = 5 batches per epoch, the last one short
    assert json.loads(f.root.attrs["training_config"])["optimizer_config"]["config"]["learning_rate"] == float(np.float32(0.01))
    assert any(not np.array_equal(a, c) for (a, _), (c, _) in zip(last, h.best_weights)) or h.best_epoch == len(val) - 1


def test_fit_matches_the_oracle_loop_on_the_same_permutations():
    topo = Topology.deployed(3)
    x, lab, _ = modulated_frames(700, seed=3)
    y = to_onehot(lab, 3)
    w = synthetic_weights(topo, seed=2)
    rng = np.random.default_rng(4)
    perms = [rng.permutation(500) for _ in range(6)]
    tr = Trainer(topo, w, device=0, lr=5e-3)
    h = tr.fit(x[:500], y[:500], batch_size=128, epochs=6, validation_data=(x[500:], y[500:]), patience=None,
               permutations=lambda ep: perms[ep])
    ref = T.fit("deployed", w, x[:500].astype(np.float64), y[:500].astype(np.float64), 128, 6, (x[500:].astype(np.float64), y[500:].astype(np.float64)),
                patience=None, permutations=lambda ep: perms[ep], dtype=np.float64, adam=dict(lr=5e-3))
    np.testing.assert_allclose(h.history["loss"], ref["loss"], rtol=2e-5)
    np.testing.assert_allclose(h.history["val_loss"], ref["val_loss"], rtol=2e-5)
    assert h.best_epoch == ref["best_epoch"]


def test_early_stopping_counts_epochs_without_improvement():
    topo = Topology.deployed(3)
    x, lab, _ = modulated_frames(400, seed=8)
    y = to_onehot(lab, 3)
    tr = Trainer(topo, synthetic_weights(topo, seed=2), device=0, lr=0.05)      # large steps: val_loss turns around early
    h = tr.fit(x[:256], y[:256], batch_size=64, epochs=60, validation_data=(x[256:], y[256:]), patience=2, seed=1)
    v = h.history["val_loss"]
    be = int(np.argmin(v))
    assert h.best_epoch == be
    assert h.stopped_epoch is not None and h.stopped_epoch == be + 2 == len(v) - 1
    assert all(val >= v[be] for val in v[be + 1:])


def test_trained_weights_flow_into_the_reference_pipeline(tmp_path):
    """(e): fit -> .h5 -> load -> float2fix text table (CNN.ipynb cell 23-25's exporter) -> reload -> the FPGA-arithmetic
    forward: the labels of the Q6.12 path agree with the float path's on the validation frames (weights quantised to
    2^-12 move few decisions)."""
    topo = Topology.deployed(3)
    x, lab = _leveled(3000, seed=41)
    m = VTCNN2.synthetic(topo, seed=3, device=0)
    m.compile(lr=0.01)
    ck = str(tmp_path / "t.wts.h5")
    m.fit(x[:2000], to_onehot(lab[:2000], 3), batch_size=1024, epochs=20, validation_data=(x[2000:], to_onehot(lab[2000:], 3)), checkpoint=ck, seed=0)
    m.load_weights(ck)
    (ckk, cb), (dk, db) = m.get_weights()
    txt = str(tmp_path / "weights.txt")
    with open(txt, "w") as fh:
        fh.write(q612.dump_weights_f3(q612.DeployedWeights(3, ckk, cb, dk, db)))
    back = q612.load_weights_f3(txt, strict=True)
    assert np.abs(back.dense_kernel - dk).max() <= 2.0 ** -12 and not any(back.negzero.values())
    mq = VTCNN2.from_txt(txt, device=0)
    _dense, lq = mq.predict_q612(x[2000:])
    lf = m.predict_classes(x[2000:])
    assert len(set(lf.tolist())) >= 2                       # a trained net, not a constant one
    assert (lq == lf).mean() >= 0.97


def test_graph_capture_of_an_epoch_replays_with_the_device_side_step_count():
    import torch
    topo = Topology.deployed(3)
    w = synthetic_weights(topo, seed=5)
    x, y = _data(topo, 2048, seed=6)
    a, b = Trainer(topo, w, device=0), Trainer(topo, w, device=0)
    xd, yd = a._frames(x), a._targets(y, 2048)
    for _ in range(3):                                   # eager: 3 epochs x 2 batches
        for s in (0, 1024):
            a.train_batch(xd, yd, None, s, 1024)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        b.read()                                          # (touch the trainer outside the capture)
        with torch.cuda.graph(g, stream=side):
            for s in (0, 1024):
                b.train_batch(xd, yd, None, s, 1024)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ra, rb = a.read(), b.read()
    assert ra["iterations"] == rb["iterations"] == 6
    assert ra["train_loss_sum"] == rb["train_loss_sum"]
    for (k, b_), (k2, b2) in zip(a.get_weights(), b.get_weights()):
        assert np.array_equal(k, k2) and np.array_equal(b_, b2)


def test_resume_from_a_full_model_checkpoint(tmp_path):
    """Adam's m, v and iter survive the .h5: training 4 + 4 steps through a save/load equals 8 steps straight."""
    topo = Topology.cnnpy(10, 10, 5)
    w = synthetic_weights(topo, seed=8, bias_scale=0.05)
    x, y = _data(topo, 512, seed=12)
    a = Trainer(topo, w, device=0)
    xd, yd = a._frames(x), a._targets(y, 512)
    for _ in range(8):
        a.train_batch(xd, yd)
    b = Trainer(topo, w, device=0)
    for _ in range(4):
        b.train_batch(xd, yd)
    path = str(tmp_path / "resume.h5")
    b.save(path)
    ck = load_keras_h5(path)
    f = H5File(path)
    weighted = [n for n in ck.layer_names if ck.weights[n]]
    rd = lambda p: f.read("optimizer_weights/Adam/" + p)
    c = Trainer(topo, [(ck.weights[n][0][1], ck.weights[n][1][1]) for n in weighted], device=0)
    c.set_optimizer_state({"iterations": int(rd("iter:0")), "m": [(rd(f"{l}/kernel/m:0"), rd(f"{l}/bias/m:0")) for l in weighted],
                           "v": [(rd(f"{l}/kernel/v:0"), rd(f"{l}/bias/v:0")) for l in weighted]})
    for _ in range(4):
        c.train_batch(xd, yd)
    assert c.read()["iterations"] == 8
    for (k, b_), (k2, b2) in zip(a.get_weights(), c.get_weights()):
        assert np.array_equal(k, k2) and np.array_equal(b_, b2)


def test_refusals():
    with pytest.raises(ValueError):
        Trainer(Topology.vtcnn2(11), synthetic_weights(Topology.vtcnn2(11)), device=0)      # T3's training is out of scope
    topo = Topology.deployed(3)
    tr = Trainer(topo, synthetic_weights(topo), device=0)
    x, y = _data(topo, 8, seed=1)
    with pytest.raises(ValueError):
        tr.evaluate(x, y[:, :2])
    with pytest.raises(ValueError):
        tr.evaluate(x[:, :1], y)
    with pytest.raises(ValueError):
        tr.train_batch(tr._frames(x), tr._targets(y, 8), None, 4, 8)
    import torch
    xd, yd = tr._frames(x), tr._targets(y, 8)
    for bad in ((torch.from_numpy(x), yd, None),                              # host memory: its data_ptr() must never reach a kernel
                (xd.double(), yd, None), (xd[:, :, ::2], yd, None), (xd, yd[:4], None),
                (xd, yd, torch.arange(8, device="cuda")),                      # int64 indices
                (xd, yd, torch.arange(8, dtype=torch.int32))):                # indices in host memory
        with pytest.raises(ValueError):
            tr.train_batch(*bad)
    with pytest.raises(ValueError):
        tr.evaluate_enqueue(torch.from_numpy(x), yd)
    with pytest.raises(ValueError):                                           # a "permutation" that points outside the data set
        tr.fit(x, y, batch_size=4, epochs=1, permutations=lambda ep: np.arange(1, 9))
    L = _cabi.lib()
    import ctypes as C
    h = C.c_void_p()
    t3 = _cabi.MdcTopology(_cabi.KIND_VTCNN2, 256, 256, 11, (C.c_int32 * 4)(0, 0, 0, 0))
    assert L.mdc_trainer_create(C.byref(t3), 0, C.byref(h)) == -95 and b"MDC_KIND_DEPLOYED" in L.mdc_last_error()
    t1 = _cabi.MdcTopology(_cabi.KIND_DEPLOYED, 3, 0, 3, (C.c_int32 * 4)(0, 0, 0, 0))
    assert L.mdc_trainer_create(C.byref(t1), 0, C.byref(h)) == 0
    try:      # a batch before the weights are set is a state error, not a launch on zeros
        assert L.mdc_train_batch(h, tr._frames(x).data_ptr(), tr._targets(y, 8).data_ptr(), 8, None, 0, 8, 1, None) == -1
        assert L.mdc_trainer_set_adam(h, -1.0, 0.9, 0.999, 1e-7) == -22
    finally:
        L.mdc_trainer_destroy(h)
    m = VTCNN2.synthetic(topo, device=0)
    with pytest.raises(ValueError):
        m.compile(loss="mse")


def test_example_train_like_cnn_py(tmp_path):
    """examples/train_like_cnn_py.py: cnn.py:42-153 call for call -- split, compile, fit with the two callbacks, load_weights
    of the checkpoint, evaluate -- and the score it prints is the checkpoint's val_loss."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_like_cnn_py", os.path.join(root, "examples", "train_like_cnn_py.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    ck = str(tmp_path / "convmodrecnets_CNN2_0.5.wts.h5")
    model, history, score, (X_test, y_test) = ex.main([ck, "--epochs", "8", "--lr", "0.01"])
    assert X_test.shape == (8100, 2, 128)                                   # 70 / 30 of 27,000: CNN.ipynb cell 5's 18,900 training frames
    assert abs(score - min(history.history["val_loss"])) <= 1e-5 * score
    assert history.history["loss"][-1] < history.history["loss"][0]
    assert model.accuracy(X_test, y_test) > 0.45
    assert load_keras_h5(ck).keras_version == "2.4.0"


@pytest.mark.parametrize("kind,topo", CASES[:3], ids=IDS[:3])
def test_gradient_is_additive_over_the_batch_at_full_size(kind, topo):
    """A size-independent property at a size the oracle would take minutes for: the gradient of the mean loss over 2^16 frames
    (1,024 waves, 64 or 32 frames each) is the frame-weighted mean of the gradients of its two unequal parts, and so is the loss."""
    n, cut = 1 << 16, 20001
    w = synthetic_weights(topo, seed=13, bias_scale=0.05)
    x, y = _data(topo, n, seed=99)
    tr = Trainer(topo, w, device=0)
    xd, yd = tr._frames(x), tr._targets(y, n)

    def part(first, count):
        tr.read(reset=True)
        tr.train_batch(xd, yd, None, first, count, apply=False)
        r = tr.read(reset=True)
        return r["train_loss_sum"], tr.gradients()

    l_all, g_all = part(0, n)
    l_a, g_a = part(0, cut)
    l_b, g_b = part(cut, n - cut)
    assert abs(l_all - (l_a + l_b)) <= 1e-9 * abs(l_all)                       # f64 sums of the same per-frame f32 losses
    for (ka, ba), (kb, bb), (k, b) in zip(g_a, g_b, g_all):
        mix_k = (ka.astype(np.float64) * cut + kb.astype(np.float64) * (n - cut)) / n
        mix_b = (ba.astype(np.float64) * cut + bb.astype(np.float64) * (n - cut)) / n
        assert _rel(k, mix_k) <= 2e-5 and _rel(b, mix_b) <= 2e-5, (_rel(k, mix_k), _rel(b, mix_b))


# ---------------------------------------------------------------------------------------------------------------------
# The OPTIONAL Dropout (mdc_trainer_set_dropout): off in every test above -- the reference's nets contain no Dropout layer.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,topo", CASES[:4], ids=IDS[:4])
def test_dropout_masks_are_the_stated_generators(kind, topo):
    """With Dropout on, the gradients of a batch addressed through a shuffle equal the oracle's with the masks of the STATED
    generator (seed, step = Adam's iteration count, the frames' data-set indices): the kernels draw exactly those bits.  The
    mask acts (loss differs from the dropout-free one), is the same until a step is applied, and evaluation never sees it."""
    import torch
    n = 700
    w = synthetic_weights(topo, seed=3, bias_scale=0.05)
    x, y = _data(topo, n, seed=17)
    perm = np.random.default_rng(5).permutation(n).astype(np.int32)
    sel = perm[100:100 + 513]
    tr = Trainer(topo, w, device=0, dropout=0.5, dropout_seed=2016)
    xd, yd, od = tr._frames(x), tr._targets(y, n), torch.from_numpy(perm).cuda()

    def grads_now():
        tr.read(reset=True)
        tr.train_batch(xd, yd, od, 100, 513, apply=False)
        r = tr.read(reset=True)
        return r["train_loss_sum"] / 513, tr.gradients()

    loss, g = grads_now()
    ref_loss, _li, ref_g, _p = T.loss_and_grads(kind, x[sel], y[sel], w, np.float64, dropout=dict(rate=0.5, seed=2016, step=0, frames=sel))
    _pl, _pli, plain_g, _pp = T.loss_and_grads(kind, x[sel], y[sel], w, np.float64)
    assert abs(loss - ref_loss) <= 1e-5 * ref_loss
    assert _rel(ref_g[-1][0], plain_g[-1][0]) > 0.1             # the mask acts: these are not the dropout-free gradients
    for (gk, gb), (rk, rb) in zip(g, ref_g):
        assert _rel(gk, rk) <= 1e-5 and _rel(gb, rb) <= 2e-5, (_rel(gk, rk), _rel(gb, rb))
    loss2, g2 = grads_now()                                     # nothing applied: the same step, the same masks, the same bits
    assert loss2 == loss and all(np.array_equal(a, c) and np.array_equal(b, d) for (a, b), (c, d) in zip(g, g2))
    ev = tr.evaluate(x, y)                                      # inference: no mask
    assert abs(ev - T.evaluate(kind, x, y, w, np.float64)) <= 1e-5 * ev


@pytest.mark.parametrize("kind,topo", [CASES[0], CASES[2]], ids=[IDS[0], IDS[2]])
def test_dropout_trajectory_draws_a_new_mask_every_step_also_under_graph_replay(kind, topo):
    import torch
    w = synthetic_weights(topo, seed=6, bias_scale=0.02)
    x, y = _data(topo, 1024, seed=23)
    frames = np.arange(1024)
    wo = [(k.copy(), b.copy()) for k, b in w]
    opt = T.KerasAdam([t.shape for t in T.flatten_weights(wo)])
    for _ in range(6):
        T.train_step(kind, x, y, wo, opt, np.float64, dropout=dict(rate=0.3, seed=9, frames=frames))
    eager, graph = (Trainer(topo, w, device=0, dropout=0.3, dropout_seed=9) for _ in range(2))
    xd, yd = eager._frames(x), eager._targets(y, 1024)
    for _ in range(6):
        eager.train_batch(xd, yd)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        graph.read()
        with torch.cuda.graph(g, stream=side):
            graph.train_batch(xd, yd)
            graph.train_batch(xd, yd)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert eager.read()["iterations"] == graph.read()["iterations"] == 6
    for (k, b), (k2, b2), (rk, rb) in zip(eager.get_weights(), graph.get_weights(), wo):
        assert np.array_equal(k, k2) and np.array_equal(b, b2)                  # the step count lives on the device
        assert np.abs(k - rk).max() <= 1e-4 and np.abs(b - rb).max() <= 1e-4    # ... and keys the oracle's masks


def test_dropout_through_the_mirror_and_its_refusals():
    topo = Topology.deployed(3)
    x, lab = _leveled(2000, seed=3)
    m = VTCNN2.synthetic(topo, seed=4, device=0)
    m.compile(lr=0.01, dropout=0.5, dropout_seed=1)
    h = m.fit(x[:1500], to_onehot(lab[:1500], 3), batch_size=512, epochs=6, validation_data=(x[1500:], to_onehot(lab[1500:], 3)), seed=0)
    assert h.history["loss"][-1] < h.history["loss"][0]
    assert m.trainer().dropout == 0.5
    L = _cabi.lib()
    assert L.mdc_trainer_set_dropout(m.trainer()._h, 1.0, 0) == -22 and L.mdc_trainer_set_dropout(m.trainer()._h, -0.1, 0) == -22
    assert L.mdc_trainer_set_dropout(None, 0.5, 0) == -22


def test_first_epoch_of_a_fresh_cnn_py_model_starts_where_the_recorded_run_did():
    """The one training run the reference recorded (cnn.ipynb cell 6's output: cnn.py's net, 5 classes, 22,500 frames, 22 batches
    of 1,024 per epoch) begins `Epoch 1/150 - loss: 1.6028 - val_loss: 1.5949`: a freshly initialised net (glorot_uniform
    conv, he_normal dense, zero biases) on frames of the data set's scale puts its logits near zero, so the first epoch's
    running loss sits just under ln 5 = 1.6094.  The data set is not here, so this is an anchor, not parity: the same
    definition on frames of that scale must start within 0.02 of the recorded figure, in 22 steps, the last batch short."""
    import math
    topo = Topology.cnnpy(10, 10, 5)
    n = 22500
    from modulationdetectioncnn_amd import synthetic_frames
    x = synthetic_frames(n + 2000, seed=7, sigma=7e-3)                    # RML2016.10a frames are unit-energy: |x| ~ 1e-2
    lab = np.random.default_rng(7).integers(0, 5, n + 2000)
    m = VTCNN2.synthetic(topo, seed=2016, device=0)
    m.compile(loss='categorical_crossentropy', optimizer='adam')
    h = m.fit(x[:n], to_onehot(lab[:n], 5), batch_size=1024, epochs=1, validation_data=(x[n:], to_onehot(lab[n:], 5)), seed=0)
    assert m.trainer().read(reset=False)["iterations"] == 22            # `22/22` in the recorded log
    assert abs(h.history["loss"][0] - 1.6028) < 0.02 and abs(h.history["val_loss"][0] - math.log(5)) < 0.02


@pytest.mark.parametrize("kind,topo", CASES[:3], ids=IDS[:3])
def test_a_shuffle_index_outside_the_data_set_is_an_error_at_read_not_a_fault(kind, topo):
    """order_dev's VALUES are read on the device: one that names no frame of the buffers is skipped there (it adds nothing to loss
    or gradient) and counted, and the epoch's read raises with the count.  The remaining frames' gradient is exactly the batch's
    without them; after the reset the trainer carries on."""
    import torch
    n = 300
    w = synthetic_weights(topo, seed=3, bias_scale=0.05)
    x, y = _data(topo, n, seed=17)
    tr = Trainer(topo, w, device=0)
    xd, yd = tr._frames(x), tr._targets(y, n)
    order = np.arange(n, dtype=np.int32)
    bad = order.copy()
    bad[[5, 77, 290]] = [n, -1, 2 ** 31 - 1]                      # one past the end, negative, far away
    tr.read(reset=True)
    tr.train_batch(xd, yd, torch.from_numpy(bad).cuda(), 0, n, apply=False)
    with pytest.raises(RuntimeError, match="3 batch positions"):
        tr.read(reset=True)
    g_bad = tr.gradients()
    keep = np.setdiff1d(order, [5, 77, 290])
    tr.train_batch(xd, yd, torch.from_numpy(keep.astype(np.int32)).cuda(), 0, keep.size, apply=False)
    r = tr.read(reset=True)                                        # clean again
    g_ok = tr.gradients()
    for (a, b), (c, d) in zip(g_bad, g_ok):                        # mean over 300 against mean over 297 of the same 297 frames
        assert _rel(a * (n / keep.size), c) <= 2e-5 and _rel(b * (n / keep.size), d) <= 2e-5
    assert r["train_frames"] == keep.size
    L = _cabi.lib()                                                # without a shuffle the host sees the range itself
    assert L.mdc_train_batch(tr._h, xd.data_ptr(), yd.data_ptr(), n, None, n - 4, 8, 0, None) == -22
    assert L.mdc_trainer_evaluate(tr._h, xd.data_ptr(), yd.data_ptr(), n, None, 0, n + 1, None) == -22
    assert L.mdc_train_batch(tr._h, xd.data_ptr(), yd.data_ptr(), -1, None, 0, 0, 0, None) == -22
