"""The training oracle (oracle/oracle_train.py) held against torch-CPU autograd and independent statements of Adam.

Nothing in the reference pins a training result (no dataset, no recorded run: SURVEY.md section 4), so the oracle's
gradients are checked against automatic differentiation of the SAME loss written with torch library ops, and its Adam
against torch.optim.Adam where the two coincide (eps = 0) and against a scalar re-derivation of TensorFlow 2.4's
update where they do not (Keras adds eps to sqrt(v), torch to sqrt(v / (1 - beta2^t)))."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fnn

from modulationdetectioncnn_amd import Topology, synthetic_weights
from oracle import oracle_np as O
from oracle import oracle_train as T
from tests.signals import modulated_frames


def _torch_loss(kind, x, y, tw):
    """Keras' forward + categorical_crossentropy on probabilities, in torch f64 library ops."""
    n = x.shape[0]
    if kind == "deployed":
        (ck, cb), (wd, bd) = tw
        F = ck.shape[-1]
        xin = x.reshape(n, 1, 2, 128)
        w = ck.permute(3, 2, 0, 1)                                     # HWIO -> OIHW
        a = torch.relu(Fnn.conv2d(Fnn.pad(xin, (1, 1)), w, cb))        # (n,F,2,129)
        flat = a.permute(0, 2, 3, 1).reshape(n, 258 * F)               # channels_last Flatten
        out = torch.relu(flat @ wd + bd)
    else:
        (ck, cb), (w1, b1), (w2, b2) = tw
        F = ck.shape[-1]
        xin = x.reshape(n, 1, 2, 128).permute(0, 3, 1, 2)              # (H,W,C)=(1,2,128) -> NCHW (n,128,1,2)
        w = ck.permute(3, 2, 0, 1)
        a = torch.relu(Fnn.conv2d(Fnn.pad(xin, (1, 1)), w, cb))        # (n,F,1,3)
        flat = a.permute(0, 2, 3, 1).reshape(n, 3 * F)
        out = torch.relu(flat @ w1 + b1) @ w2 + b2
    p = torch.softmax(out, dim=-1)
    q = p / p.sum(dim=-1, keepdim=True)
    q = torch.clamp(q, 1e-7, 1 - 1e-7)
    li = -(y * torch.log(q)).sum(dim=-1)
    return li.mean(), li, p


def _case(kind, F, seed, n=96, classes=None):
    topo = Topology.deployed(F, 3) if kind == "deployed" else Topology.cnnpy(F, 10, classes or 5)
    w = synthetic_weights(topo, seed=seed, bias_scale=0.05)
    x, lab, _ = modulated_frames(n, seed=seed)
    if kind == "cnnpy":
        x = x * 40.0                     # T4's random-init logits are otherwise ~0: give the ReLUs something to cut
    y = T.onehot(lab % topo.classes, topo.classes, np.float64)
    return topo, w, x.astype(np.float64), y


@pytest.mark.parametrize("kind,F", [("deployed", 3), ("deployed", 10), ("cnnpy", 10), ("cnnpy", 4)])
def test_gradients_match_torch_autograd(kind, F):
    topo, w, x, y = _case(kind, F, seed=11)
    loss, li, grads, p = T.loss_and_grads(kind, x, y, w, np.float64)
    tw = [(torch.tensor(k, dtype=torch.float64, requires_grad=True), torch.tensor(b, dtype=torch.float64, requires_grad=True)) for k, b in w]
    tl, tli, tp = _torch_loss(kind, torch.tensor(x), torch.tensor(y), tw)
    tl.backward()
    assert abs(loss - tl.item()) < 1e-12
    np.testing.assert_allclose(li, tli.detach().numpy(), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(p, tp.detach().numpy(), rtol=1e-12, atol=1e-15)
    for (dk, db), (tk, tb) in zip(grads, tw):
        for g, t in ((dk, tk), (db, tb)):
            ref = t.grad.numpy()
            assert np.abs(ref).max() > 0
            assert np.abs(g - ref).max() <= 1e-10 * np.abs(ref).max()


def test_forward_is_the_inference_oracle():
    for kind, F in (("deployed", 3), ("cnnpy", 10)):
        topo, w, x, y = _case(kind, F, seed=5)
        _l, _li, _g, p = T.loss_and_grads(kind, x, y, w, np.float64)
        np.testing.assert_allclose(p, O.forward(kind, x, w, dtype=np.float64)["probs"], rtol=1e-12, atol=1e-15)


def test_loss_is_the_evaluate_oracle():
    topo, w, x, y = _case("deployed", 3, seed=9)
    loss, _li, _g, p = T.loss_and_grads("deployed", x, y, w, np.float32)
    assert abs(loss - O.categorical_crossentropy(p, y.argmax(axis=1))) < 1e-6


def test_clip_blocks_the_gradient_outside_the_interval():
    """A frame classified wrongly with probability < 1e-7 contributes log(1e-7) to the loss and NOTHING to the gradient
    (tf.clip_by_value's gradient), as does one classified rightly with probability > 1 - 1e-7."""
    p = np.array([[1e-9, 1 - 1e-9, 0.0], [0.2, 0.5, 0.3]], np.float64)
    y = np.array([[1, 0, 0], [1, 0, 0]], np.float64)
    li, gp = T.crossentropy_on_probs(p, y)
    assert abs(li[0] + np.log(1e-7)) < 1e-12 and np.all(gp[0] == 0)
    assert abs(li[1] + np.log(0.2)) < 1e-12 and gp[1, 0] < 0
    # general targets (label smoothing): every clipped class is masked on its own
    tp, ty = torch.tensor(p, requires_grad=True), torch.tensor([[0.9, 0.05, 0.05], [0.9, 0.05, 0.05]], dtype=torch.float64)
    q = torch.clamp(tp / tp.sum(-1, keepdim=True), 1e-7, 1 - 1e-7)
    (-(ty * torch.log(q)).sum(-1)).sum().backward()
    _li, gp2 = T.crossentropy_on_probs(p, ty.numpy())
    np.testing.assert_allclose(gp2, tp.grad.numpy(), rtol=1e-10, atol=1e-12)


def test_adam_equals_torch_adam_at_eps_zero():
    rng = np.random.default_rng(3)
    shapes = [(7, 3), (3,)]
    params = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    tparams = [torch.tensor(p.copy(), requires_grad=True) for p in params]
    opt = T.KerasAdam(shapes, eps=0.0)
    topt = torch.optim.Adam(tparams, lr=1e-3, betas=(0.9, 0.999), eps=0.0)
    for _ in range(25):
        grads = [(rng.standard_normal(s) * 0.1 + 0.05).astype(np.float32) for s in shapes]
        opt.apply(params, grads)
        for tp, g in zip(tparams, grads):
            tp.grad = torch.tensor(g)
        topt.step()
    for p, tp in zip(params, tparams):
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=2e-6, atol=2e-7)


def test_adam_is_tensorflows_epsilon_hat_form():
    """Scalar re-derivation in f64: theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps) -- eps is NOT scaled by the
    bias correction (torch's and the paper's form divide v first).  A gradient of order eps makes the difference visible."""
    g_seq = [3e-7, 1e-7, 2e-7, 5e-8, 4e-7]
    th, th_paper, m, v = 0.5, 0.5, 0.0, 0.0
    opt = T.KerasAdam([(1,)])
    p = [np.array([0.5], np.float32)]
    for t, g in enumerate(g_seq, 1):
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        th -= 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-7)
        th_paper -= 1e-3 * (m / (1 - 0.9 ** t)) / (np.sqrt(v / (1 - 0.999 ** t)) + 1e-7)
        opt.apply(p, [np.array([g], np.float32)])
    assert abs(p[0][0] - th) < 2e-7
    assert abs(th - th_paper) > 1e-3 > 50 * abs(p[0][0] - th)      # the two forms are 1e-3 apart here; the oracle is TF's
    assert opt.iterations == len(g_seq)


def test_train_step_reduces_the_loss_and_f32_tracks_f64():
    topo, w, x, y = _case("deployed", 3, seed=21, n=256)
    w32 = [(k.copy(), b.copy()) for k, b in w]
    w64 = [(k.copy(), b.copy()) for k, b in w]
    o32 = T.KerasAdam([t.shape for t in T.flatten_weights(w32)])
    o64 = T.KerasAdam([t.shape for t in T.flatten_weights(w64)])
    l32 = [T.train_step("deployed", x, y, w32, o32, np.float32) for _ in range(20)]
    l64 = [T.train_step("deployed", x, y, w64, o64, np.float64) for _ in range(20)]
    assert l32[-1] < l32[0]
    np.testing.assert_allclose(l32, l64, rtol=1e-5)
    for (a, b), (c, d) in zip(w32, w64):
        assert np.abs(a - c).max() < 1e-4 and np.abs(b - d).max() < 1e-4


def test_fit_callbacks_early_stopping_and_best_checkpoint():
    """patience counts epochs WITHOUT improvement (strict <); the kept weights are those of the best epoch."""
    topo, w, x, y = _case("deployed", 3, seed=4, n=192)
    xv, yv = x[128:], y[128:]
    rng = np.random.default_rng(0)
    perms = [rng.permutation(128) for _ in range(40)]
    # a large learning rate makes the validation loss turn around quickly
    h = T.fit("deployed", w, x[:128], y[:128], batch_size=50, epochs=40, validation_data=(xv, yv), patience=3,
              permutations=lambda ep: perms[ep], adam=dict(lr=0.02))
    v = h["val_loss"]
    be = int(np.argmin(v))
    assert h["best_epoch"] == be == v.index(min(v))
    if h["stopped_epoch"] is not None:
        assert h["stopped_epoch"] == be + 3 and len(v) == be + 4
        assert all(val >= v[be] for val in v[be + 1:])
    assert abs(T.evaluate("deployed", xv, yv, h["best_weights"]) - v[be]) < 1e-7
    # 128 frames in batches of 50: 50 + 50 + 28 -- the short batch is trained on (3 Adam steps per epoch)
    assert h["opt"].iterations == 3 * len(v)


@pytest.mark.parametrize("rate", [0.1, 0.5, 0.6])
def test_dropout_generator_statistics(rate):
    """The stated counter-based generator behind the OPTIONAL Dropout (include/mdc.h, mdc_trainer_set_dropout): keep fraction
    1 - rate within 4 sigma, scale 1 / (1 - rate), a fresh mask at every step and for every site, no correlation between
    neighbouring elements or frames, and a frame's mask independent of which other frames are in the batch."""
    n, e = 4000, 774
    m = T.dropout_scale(rate, 2016, 0, np.arange(n), e, 0)
    keep = m > 0
    sigma = np.sqrt(rate * (1 - rate) / (n * e))
    assert abs(keep.mean() - (1 - rate)) < 4 * sigma
    assert np.allclose(m[keep], 1.0 / (1.0 - np.float64(np.float32(rate))))
    for other in (T.dropout_scale(rate, 2016, 1, np.arange(n), e, 0) > 0,       # next step
                  T.dropout_scale(rate, 2016, 0, np.arange(n), e, 1) > 0,       # other site
                  T.dropout_scale(rate, 2017, 0, np.arange(n), e, 0) > 0):      # other seed
        agree = (keep == other).mean()
        assert abs(agree - (rate ** 2 + (1 - rate) ** 2)) < 5e-3                # what two independent masks agree on
    assert abs(np.corrcoef(keep[:, :-1].ravel(), keep[:, 1:].ravel())[0, 1]) < 5e-3
    assert abs(np.corrcoef(keep[:-1].ravel(), keep[1:].ravel())[0, 1]) < 5e-3
    sub = T.dropout_scale(rate, 2016, 0, np.array([17, 3999, 5]), e, 0)
    assert np.array_equal(sub, m[[17, 3999, 5]])


@pytest.mark.parametrize("kind,F", [("deployed", 3), ("cnnpy", 10)])
def test_dropout_gradients_match_torch_autograd(kind, F):
    topo, w, x, y = _case(kind, F, seed=13)
    drop = dict(rate=0.5, seed=7, step=3, frames=np.arange(len(x)) + 100)
    loss, li, grads, p = T.loss_and_grads(kind, x, y, w, np.float64, dropout=drop)
    tw = [(torch.tensor(k, dtype=torch.float64, requires_grad=True), torch.tensor(b, dtype=torch.float64, requires_grad=True)) for k, b in w]
    n = len(x)
    tx, ty = torch.tensor(x), torch.tensor(y)
    if kind == "deployed":
        (ck, cb), (wd, bd) = tw
        a = torch.relu(Fnn.conv2d(Fnn.pad(tx.reshape(n, 1, 2, 128), (1, 1)), ck.permute(3, 2, 0, 1), cb))
        flat = a.permute(0, 2, 3, 1).reshape(n, 258 * F) * torch.tensor(T.dropout_scale(0.5, 7, 3, drop["frames"], 258 * F, 0))
        out = torch.relu(flat @ wd + bd)
    else:
        (ck, cb), (w1, b1), (w2, b2) = tw
        a = torch.relu(Fnn.conv2d(Fnn.pad(tx.reshape(n, 1, 2, 128).permute(0, 3, 1, 2), (1, 1)), ck.permute(3, 2, 0, 1), cb))
        flat = a.permute(0, 2, 3, 1).reshape(n, 3 * F) * torch.tensor(T.dropout_scale(0.5, 7, 3, drop["frames"], 3 * F, 0))
        h = torch.relu(flat @ w1 + b1) * torch.tensor(T.dropout_scale(0.5, 7, 3, drop["frames"], w1.shape[1], 1))
        out = h @ w2 + b2
    pr = torch.softmax(out, dim=-1)
    q = torch.clamp(pr / pr.sum(dim=-1, keepdim=True), 1e-7, 1 - 1e-7)
    tl = (-(ty * torch.log(q)).sum(dim=-1)).mean()
    tl.backward()
    assert abs(loss - tl.item()) < 1e-12
    loss0, *_ = T.loss_and_grads(kind, x, y, w, np.float64)
    assert abs(loss - loss0) > 1e-6                                             # the mask acts
    for (dk, db), (tk, tb) in zip(grads, tw):
        for g, t in ((dk, tk), (db, tb)):
            ref = t.grad.numpy()
            assert np.abs(g - ref).max() <= 1e-10 * np.abs(ref).max()
