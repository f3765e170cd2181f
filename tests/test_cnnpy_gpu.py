"""Parity of the literal cnn.py model (cnn.py:104-115, "T4") on the dense_chain HIP kernel against
the CPU oracle.  No weights are bundled for this topology: synthetic weights, oracle-only parity
("parity unpinned").  Tolerance 2e-5 relative to the largest |logit| (exact-f32 MFMA)."""
import numpy as np
import pytest
import torch

from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames, synthetic_weights
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 15, 16, 17, 64, 1000, 20000])
@pytest.mark.parametrize("shape", [(10, 10, 5), (3, 16, 11), (7, 4, 2)])
def test_parity(n, shape):
    topo = Topology.cnnpy(*shape)
    w = synthetic_weights(topo, seed=17, bias_scale=0.05)
    x = synthetic_frames(n, seed=4, sigma=0.05)
    ref = O.forward("cnnpy", x, w, dtype=np.float64)
    m = VTCNN2(topo)
    m.set_weights(w)
    scale = float(np.abs(ref["logits"]).max())
    lg = m.predict(x, tap="dense")
    assert np.abs(lg - ref["logits"]).max() <= 2e-5 * scale
    np.testing.assert_allclose(m.predict(x), ref["probs"], atol=2e-6)
    lab = m.predict_classes(x)
    srt = np.sort(ref["logits"], axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 1e-4 * scale
    assert (lab[decided] == ref["labels"][decided]).all()
    assert decided.mean() > 0.99
    np.testing.assert_allclose(m.predict(x, tap="hidden"), ref["dense1"], atol=2e-5 * max(1.0, np.abs(ref["dense1"]).max()))
    conv = m.predict(x, tap="conv")
    assert conv.shape == (n, 1, 3, shape[0])
    np.testing.assert_allclose(conv.reshape(n, -1), ref["flat"], atol=2e-5 * max(1.0, np.abs(ref["flat"]).max()))


def test_default_cnnpy_shape_and_ties():
    m = VTCNN2.synthetic("cnnpy", classes=5)          # cnn.py:47: five classes, F=10, D=10
    assert m.topology == Topology.cnnpy(10, 10, 5)
    x = np.zeros((33, 2, 128), np.float32)            # zero input, zero biases -> all logits 0 -> uniform, label 0
    np.testing.assert_array_equal(m.predict(x), np.full((33, 5), np.float32(0.2)))
    assert (m.predict_classes(x) == 0).all()
    xt = synthetic_frames(3000, seed=1, device="cuda")
    a = m.predict(xt)
    assert torch.equal(a, m.predict(xt, batch_size=777))
