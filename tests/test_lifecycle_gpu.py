"""Model life-cycle stress: seeded random SEQUENCES of the mirror's calls -- set_weights, predict at ragged sizes (numpy and
device tensors, taps), compile / fit, load_weights, save / from_h5, a second model interleaved on another stream -- with the
numpy oracle tracking what the weights must be after every step.  What it is for: state that outlives a call (the packed
engine, its workspaces, the trainer and its optimizer slots) going stale when calls arrive in an order no other test uses."""
import os

import numpy as np
import pytest
import torch

from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames, synthetic_weights
from modulationdetectioncnn_amd.training import to_onehot
from oracle import oracle_np as O
from oracle import oracle_train as T

pytestmark = pytest.mark.gpu

TOPOS = [("deployed", Topology.deployed(3)), ("deployed", Topology.deployed(10)), ("cnnpy", Topology.cnnpy(10, 10, 5))]


def _check(m, kind, w, rng, tol=3e-5):
    n = int(rng.choice([1, 2, 17, 64, 65, 300, 1025]))
    x = synthetic_frames(n, seed=int(rng.integers(1 << 30)), sigma=0.05 if kind == "deployed" else 0.5)
    ref = O.forward(kind, x, w, dtype=np.float64)
    how = int(rng.integers(3))
    if how == 0:
        p = m.predict(x, batch_size=int(rng.choice([32, 1024])))
    elif how == 1:
        p = m.predict(torch.from_numpy(x).cuda()).cpu().numpy()
    else:
        p = m.predict(x)
        d = m.predict(x, tap="dense")
        key = "dense" if kind == "deployed" else "logits"
        assert np.abs(d - ref[key]).max() <= tol * max(1.0, np.abs(ref[key]).max())
    assert p.shape == ref["probs"].shape and np.abs(p - ref["probs"]).max() <= tol
    margin = np.sort(ref["probs"], axis=1)[:, -1] - np.sort(ref["probs"], axis=1)[:, -2]
    lab = m.predict_classes(x)
    assert (lab[margin > 1e-4] == ref["labels"][margin > 1e-4]).all()


@pytest.mark.parametrize("seed", range(6))
def test_random_call_sequences_keep_the_model_consistent(tmp_path, seed):
    rng = np.random.default_rng(100 + seed)
    kind, topo = TOPOS[seed % len(TOPOS)]
    w = synthetic_weights(topo, seed=seed, bias_scale=0.05)
    m = VTCNN2(topo, device=0)
    m.set_weights(w)
    w = [(k.copy(), b.copy()) for k, b in w]
    other = VTCNN2.synthetic(TOPOS[(seed + 1) % len(TOPOS)][1], seed=99, device=0)      # a second model sharing the device
    w_other = other.get_weights()
    side = torch.cuda.Stream()
    opt = None                                     # the oracle's Adam, alive as long as the mirror's trainer is
    n_tr = 700
    xt = synthetic_frames(n_tr, seed=5, sigma=0.05 if kind == "deployed" else 0.5)
    yt = to_onehot(np.random.default_rng(5).integers(0, topo.classes, n_tr), topo.classes)
    for step in range(14):
        op = int(rng.integers(7))
        if op == 0:                                # new weights from outside (keeps the optimizer's slots, as Keras does)
            w = synthetic_weights(topo, seed=int(rng.integers(1 << 20)), bias_scale=0.05)
            m.set_weights(w)
        elif op == 1:                              # one epoch of fit, no shuffle: the oracle runs the same steps
            if opt is None:
                m.compile(loss="categorical_crossentropy", optimizer="adam", lr=2e-3)
                opt = T.KerasAdam([t.shape for t in T.flatten_weights(w)], lr=2e-3)
            m.fit(xt, yt, batch_size=256, epochs=1, shuffle=False, patience=None)
            w = [(k.astype(np.float32).copy(), b.astype(np.float32).copy()) for k, b in w]
            for s in range(0, n_tr, 256):
                T.train_step(kind, xt[s:s + 256], yt[s:s + 256], w, opt, np.float64)
            got = m.get_weights()
            for (k, b), (gk, gb) in zip(w, got):
                assert np.abs(gk - k).max() <= 2e-4 and np.abs(gb - b).max() <= 2e-4, step
            w = [(gk.copy(), gb.copy()) for gk, gb in got]          # follow the device's bits from here (f32 against f64 drift)
            opt_state = m.trainer().optimizer_state()
            assert opt_state["iterations"] == opt.iterations, (opt_state["iterations"], opt.iterations)
            opt.m = [a.copy() for pair in opt_state["m"] for a in pair]
            opt.v = [a.copy() for pair in opt_state["v"] for a in pair]
        elif op == 2:                              # save -> a fresh model from the file predicts the same bits
            path = str(tmp_path / f"s{step}.h5")
            m.save(path)
            m2 = VTCNN2.from_h5(path, device=0)
            x = synthetic_frames(33, seed=step)
            assert np.array_equal(m2.predict(x), m.predict(x))
        elif op == 3:                              # load_weights of an earlier save into THIS model
            path = str(tmp_path / f"l{step}.h5")
            VTCNN2.synthetic(topo, seed=step, device=0).save(path)
            m.load_weights(path)
            w = VTCNN2.from_h5(path, device=0).get_weights()
        elif op == 4:                              # the other model runs on another stream in between
            with torch.cuda.stream(side):
                xo = torch.from_numpy(synthetic_frames(257, seed=step)).cuda()
                po = other.predict(xo)
            side.synchronize()
            k_other = TOPOS[(seed + 1) % len(TOPOS)][0]
            ro = O.forward(k_other, xo.cpu().numpy(), w_other, dtype=np.float64)
            assert np.abs(po.cpu().numpy() - ro["probs"]).max() <= 3e-5
        elif op == 5:                              # a compile in the middle resets the optimizer, not the weights
            m.compile(loss="categorical_crossentropy", optimizer="adam", lr=2e-3)
            opt = T.KerasAdam([t.shape for t in T.flatten_weights(w)], lr=2e-3)
        _check(m, kind, w, rng)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "fp8"])
def test_a_reused_vtcnn2_model_equals_a_fresh_one(dtype):
    """The canonical VT-CNN2's engine packs its weights for the dtype (bf16 hi/lo splits, E4M3 with scales derived from the weights or
    from a calibration): after any sequence of set_weights / calibrate / predict at other sizes, a model must give the bits a
    freshly built one with the same weights and settings gives."""
    topo = Topology.vtcnn2(11)
    rng = np.random.default_rng(7)
    x = synthetic_frames(300, seed=3)
    m = VTCNN2(topo, device=0, dtype=dtype)
    for round_ in range(3):
        w = synthetic_weights(topo, seed=50 + round_)
        m.set_weights(w)
        for n in rng.choice([1, 16, 17, 255, 300], size=2):
            m.predict(x[:int(n)])                                      # other sizes first: small-batch forms, other workspaces
        absmax = None
        if dtype == "fp8" and round_ != 1:
            absmax = m.calibrate_fp8_features(x[:64])
            m.predict(x[:5])
        fresh = VTCNN2(topo, device=0, dtype=dtype, **({"fp8_feature_absmax": absmax} if absmax is not None else {}))
        fresh.set_weights(w)
        if dtype == "fp8" and round_ == 1:
            m.fp8_feature_absmax = None                                # back to the statistical estimate from the weights
        a, b = m.predict(x), fresh.predict(x)
        assert np.array_equal(a, b), (dtype, round_, np.abs(a - b).max())
        assert np.array_equal(m.predict_classes(x), fresh.predict_classes(x))


def test_what_the_engine_was_built_from_cannot_change_under_it():
    topo = Topology.vtcnn2(3)
    x = synthetic_frames(64, seed=2, sigma=0.004)
    m = VTCNN2.synthetic(topo, seed=1, device=0, dtype="fp8")
    a = m.predict(x)
    for name, value in (("dtype", "bf16"), ("topology", Topology.deployed(3)), ("fp8_bf16_features", True)):
        with pytest.raises(AttributeError, match="fixed at construction"):
            setattr(m, name, value)
    m.fp8_input_absmax = 0.5                                           # allowed: a coarser activation scale, re-packed at the next use
    b = m.predict(x)
    fresh = VTCNN2.synthetic(topo, seed=1, device=0, dtype="fp8", fp8_input_absmax=0.5)
    assert np.array_equal(b, fresh.predict(x)) and not np.array_equal(a, b)
