"""Label-agreement floors of the reduced-precision modes against the exact-f32 kernels (cnn.py:209: the label is
`int(np.argmax(test_Y_hat[i,:]))`).

The margin tests (test_vtcnn2_gpu.py::_check, test_deployed_gpu.py) only compare labels where the oracle's top-2
margin exceeds a multiple of the mode's tolerance, which for fp8 is a third of the largest logit: a regression that
flipped a tenth of the labels could pass them.  Here every frame counts: on >= 2^16 synthetic frames the fraction of
frames whose label equals the f32 kernels' label must stay above the floor DESIGN.md quotes for the mode.  The f32
kernels themselves are the ones pinned to the f64 oracle (and, for T1, to Keras' recorded output); north_star's
"argmax bit-exact" holds for f32 only, and these floors say how far the narrower modes are from it.
Parity unpinned for T3 (no reference weights exist): synthetic seed-2016 weights, N(0, 5e-3) frames."""
import numpy as np
import pytest
import torch

from conftest import load_deployed_npz
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames, synthetic_weights

pytestmark = pytest.mark.gpu

N = 1 << 16


def _agreement(m_ref, m, x):
    a = m_ref.predict_classes(x)
    b = m.predict_classes(x)
    return float((a == b).float().mean())


@pytest.mark.parametrize("dtype,floor", [("bf16", 0.998), ("fp8", 0.985)])
@pytest.mark.parametrize("classes", [11, 3])
def test_vtcnn2_label_agreement_floor(dtype, floor, classes):
    topo = Topology.vtcnn2(classes)
    w = synthetic_weights(topo, seed=2016)
    mf = VTCNN2(topo, dtype="f32")
    mf.set_weights(w)
    m = VTCNN2(topo, dtype=dtype)
    m.set_weights(w)
    x = synthetic_frames(N, seed=2016, device="cuda")
    agree = _agreement(mf, m, x)
    assert agree >= floor, f"vtcnn2 C={classes} {dtype}: {agree:.5f} of labels equal the f32 kernels' (floor {floor})"


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("name", ["3convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5", "5convmodrecnets_CNN2_0.5"])
@pytest.mark.parametrize("sigma", [5e-3, 0.1])
def test_deployed_label_agreement_floor(dtype, name, sigma):
    """sigma 5e-3 is the bundled frames' scale; at 0.1 the class sums are far from the bias-dominated regime and
    the three classes all occur."""
    w = load_deployed_npz(name)
    topo = Topology.deployed(w[0][1].shape[0], 3)
    mf = VTCNN2(topo, dtype="f32")
    mf.set_weights(w)
    m = VTCNN2(topo, dtype=dtype)
    m.set_weights(w)
    x = synthetic_frames(N, seed=77, sigma=sigma, device="cuda")
    agree = _agreement(mf, m, x)
    assert agree >= 0.999, f"{name} {dtype} sigma={sigma}: {agree:.5f} of labels equal the f32 kernel's (floor 0.999)"
