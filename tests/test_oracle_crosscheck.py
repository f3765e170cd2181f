"""Second, independent statement of each topology in torch-CPU (F.conv2d + matmul), checked
against the numpy oracle.  T2/T3/T4 have no recorded reference outputs ("parity unpinned");
this is what guards their restatement.  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from modulationdetectioncnn_amd.topology import Topology, synthetic_frames, synthetic_weights
from oracle import oracle_np as O


def torch_deployed(x, w):
    (ck, cb), (dk, db) = [(torch.from_numpy(k).double(), torch.from_numpy(b).double()) for k, b in w]
    t = torch.from_numpy(x).double()[:, None]                      # NCHW with C=1, H=2, W=128
    t = F.pad(t, (1, 1))                                           # ZeroPadding2D((0,1))
    y = F.relu(F.conv2d(t, ck.permute(3, 2, 0, 1), cb))            # HWIO -> OIHW; (n,F,2,129)
    flat = y.permute(0, 2, 3, 1).reshape(x.shape[0], -1)           # channels_last flatten
    return F.relu(flat @ dk + db)


def torch_vtcnn2(x, w):
    (k1, b1), (k2, b2), (w1, c1), (w2, c2) = [(torch.from_numpy(k).double(), torch.from_numpy(b).double()) for k, b in w]
    t = F.pad(torch.from_numpy(x).double()[:, None], (2, 2))
    t = F.relu(F.conv2d(t, k1, b1))
    t = F.relu(F.conv2d(F.pad(t, (2, 2)), k2, b2))
    flat = t.reshape(x.shape[0], -1)                               # channels_first flatten: o*132 + w
    return F.relu(flat @ w1 + c1) @ w2 + c2, flat


def torch_cnnpy(x, w):
    (ck, cb), (w1, c1), (w2, c2) = [(torch.from_numpy(k).double(), torch.from_numpy(b).double()) for k, b in w]
    t = torch.from_numpy(x).double().permute(0, 2, 1)[:, :, None, :]   # NHWC (n,1,2,128) -> NCHW (n,128,1,2)
    t = F.relu(F.conv2d(F.pad(t, (1, 1)), ck.permute(3, 2, 0, 1), cb))  # (n,F,1,3)
    flat = t.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    return F.relu(flat @ w1 + c1) @ w2 + c2


@pytest.mark.parametrize("filters", [3, 10])
def test_deployed(filters):
    topo = Topology.deployed(filters)
    w = synthetic_weights(topo, seed=7, bias_scale=0.05)
    x = synthetic_frames(33, seed=1, sigma=0.05)
    got = O.forward("deployed", x, w, dtype=np.float64)
    np.testing.assert_allclose(got["dense"], torch_deployed(x, w).numpy(), atol=1e-12)
    assert got["conv"].shape == (33, 2, 129, filters) and got["flat"].shape == (33, 258 * filters)


@pytest.mark.parametrize("classes", [3, 11])
def test_vtcnn2(classes):
    topo = Topology.vtcnn2(classes)
    w = synthetic_weights(topo, seed=3, bias_scale=0.02)
    x = synthetic_frames(5, seed=2, sigma=0.05)
    got = O.forward("vtcnn2", x, w, dtype=np.float64, chunk=2)
    logits, flat = torch_vtcnn2(x, w)
    np.testing.assert_allclose(got["flat"], flat.numpy(), atol=1e-12)
    np.testing.assert_allclose(got["logits"], logits.numpy(), atol=1e-11)
    assert got["flat"].shape == (5, 10560) and got["probs"].shape == (5, classes)
    assert topo.flops_per_frame == {3: 38247936, 11: 38252032}[classes]


def test_cnnpy():
    topo = Topology.cnnpy()
    w = synthetic_weights(topo, seed=5, bias_scale=0.05)
    x = synthetic_frames(9, seed=4, sigma=0.05)
    got = O.forward("cnnpy", x, w, dtype=np.float64)
    np.testing.assert_allclose(got["logits"], torch_cnnpy(x, w).numpy(), atol=1e-12)
    assert got["flat"].shape == (9, 30) and got["probs"].shape == (9, 5)


def test_f32_oracle_close_to_f64():
    topo = Topology.vtcnn2(11)
    w = synthetic_weights(topo, seed=2016)
    x = synthetic_frames(4, seed=2016)
    a = O.forward("vtcnn2", x, w, dtype=np.float32)
    b = O.forward("vtcnn2", x, w, dtype=np.float64)
    scale = np.abs(b["logits"]).max()
    assert np.abs(a["logits"] - b["logits"]).max() < 2e-5 * scale


@pytest.mark.parametrize("kind", ["deployed", "vtcnn2", "cnnpy"])
def test_torch_port_matches_numpy_oracle(kind):
    """oracle_torch (bench.py's multithreaded CPU baseline) against oracle_np (the one pinned to the golden vectors)."""
    from conftest import load_deployed_npz
    from oracle import oracle_torch as OT
    from modulationdetectioncnn_amd.topology import Topology, synthetic_frames, synthetic_weights
    x = np.asarray(synthetic_frames(48, seed=11)) * (4.0 if kind == "deployed" else 1.0)
    if kind == "deployed":
        w = load_deployed_npz("convmodrecnets_CNN2_0.5")
    else:
        w = synthetic_weights(Topology.vtcnn2(11) if kind == "vtcnn2" else Topology.cnnpy(10, 10, 5), seed=2016)
    a = O.forward(kind, x, w, dtype=np.float64)
    b = OT.forward(kind, x, w)
    assert np.abs(a["probs"] - b["probs"]).max() < 2e-6
    srt = np.sort(a["probs"], axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 1e-5
    assert (a["labels"][decided] == b["labels"][decided]).all()
    assert OT.forward(kind, x[:0], w)["probs"].shape[0] == 0


def test_theano_kernel_flip_is_a_true_convolution():
    """`set_weights(..., theano_kernels=True)` (a Keras-1/Theano checkpoint of the canonical VT-CNN2): flipping the 4-D
    kernels once turns the correlation every path computes into Theano's convolution.  Checked on the host against
    scipy.signal.convolve2d for conv1 of the canonical net (no GPU needed: only the weight handling is under test)."""
    from scipy.signal import convolve2d, correlate2d
    from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_weights
    topo = Topology.vtcnn2(3)
    w = synthetic_weights(topo, seed=4)
    m = VTCNN2(topo, device=0)
    m.set_weights(w, theano_kernels=True)
    k_th = w[0][0]                                   # (256,1,1,3) as a Theano checkpoint stores it
    k_used = m.get_weights()[0][0]                   # what the correlation kernels will see
    np.testing.assert_array_equal(k_used, k_th[:, :, ::-1, ::-1])
    x = np.random.default_rng(0).standard_normal((2, 132)).astype(np.float64)
    for c in (0, 17, 255):
        np.testing.assert_allclose(correlate2d(x, k_used[c, 0].astype(np.float64), mode="valid"),
                                   convolve2d(x, k_th[c, 0].astype(np.float64), mode="valid"), atol=1e-12)
    m2 = VTCNN2(topo, device=0)
    m2.set_weights(w)                                # default: Keras-2 semantics, kernels untouched
    np.testing.assert_array_equal(m2.get_weights()[1][0], w[1][0])
    # deployed nets (HWIO): the flip is over axes (0, 1)
    d = VTCNN2(Topology.deployed(3, 3), device=0)
    wd = synthetic_weights(Topology.deployed(3, 3), seed=1)
    d.set_weights(wd, theano_kernels=True)
    np.testing.assert_array_equal(d.get_weights()[0][0], wd[0][0][::-1, ::-1])


def test_categorical_crossentropy_is_keras_formula():
    """oracle_np.categorical_crossentropy (the checker of VTCNN2.evaluate, cnn.py:153) against the formula written out, and
    against torch's NLL on the same clipped probabilities."""
    import torch
    from oracle import oracle_np as O
    rng = np.random.default_rng(3)
    z = rng.normal(0, 3, (500, 5)).astype(np.float32)
    p = O.softmax(z)
    p[0] = [1, 0, 0, 0, 0]                                   # a saturated row: the clip decides
    t = rng.integers(0, 5, size=500)
    t[0] = 1
    want = 0.0
    for i in range(500):
        q = p[i] / p[i].sum()
        want += -np.log(min(max(float(q[t[i]]), 1e-7), 1 - 1e-7))
    want /= 500
    got = O.categorical_crossentropy(p, t)
    assert got == pytest.approx(want, rel=1e-6)
    pt = torch.from_numpy(p / p.sum(axis=1, keepdims=True)).clamp(1e-7, 1 - 1e-7)
    assert got == pytest.approx(float(torch.nn.functional.nll_loss(pt.log(), torch.from_numpy(t))), rel=1e-5)
