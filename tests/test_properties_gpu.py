"""Property-based checks on the GPU (hypothesis picks the sizes, hops, chunkings): whatever the batch size, however it
is chunked, wherever a window starts in a capture and whichever entry point carries it, a frame's result is the same
bits.  Ragged tails, single frames, odd hops (2-byte aligned windows) and chunk boundaries come up by themselves."""
import functools
import os

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from modulationdetectioncnn_amd import VTCNN2, Topology, _cabi, frames_from_iq_u8, synthetic_frames

pytestmark = pytest.mark.gpu

# reproducible examples by default (the round-end run must not depend on a seed); MDC_PROP_EXAMPLES=N draws N random ones
_N = int(os.environ.get("MDC_PROP_EXAMPLES", "0"))
_S = dict(max_examples=_N, derandomize=False, deadline=None) if _N else dict(max_examples=40, derandomize=True, deadline=None)

MODELS = ["deployed3-f32", "deployed10-f32", "deployed10-bf16", "deployed3-f16", "deployed10-fp8", "vtcnn2-bf16", "vtcnn2-fp8", "vtcnn2-f32"]


@functools.lru_cache(maxsize=None)
def _model(key):
    topo, dtype = key.split("-")
    return VTCNN2.synthetic(Topology.vtcnn2(11) if topo == "vtcnn2" else topo, dtype=dtype)


def _limit(key, n):
    return min(n, 700) if key.startswith("vtcnn2") else n


@settings(**_S)
@given(st.sampled_from(MODELS), st.integers(0, 3000), st.integers(1, 200), st.integers(0, 2 ** 31 - 1))
def test_raw_byte_windows_equal_convert_then_forward(key, n, hop, seed):
    m = _model(key)
    n = _limit(key, n)
    nbytes = 0 if n == 0 else 2 * hop * (n - 1) + 256
    if hop == 128:
        nbytes = 256 * n
    iq = torch.from_numpy(np.random.default_rng(seed).integers(0, 256, size=nbytes, dtype=np.uint8)).cuda()
    scale = 0.02 / 127.5
    p, l = m.predict_iq_u8(iq, scale=scale, hop=hop)
    x = frames_from_iq_u8(iq, scale, hop=hop)
    assert x.shape[0] == n
    p2, l2, _ = m.forward_device(x)
    assert torch.equal(p, p2) and torch.equal(l, l2)


@settings(**_S)
@given(st.sampled_from(MODELS), st.integers(1, 4000), st.data())
def test_results_do_not_depend_on_the_chunking(key, n, data):
    m = _model(key)
    n = _limit(key, n)
    bs = data.draw(st.integers(1, n))
    x = synthetic_frames(n, seed=n, device="cuda")
    p, l, _ = m.forward_device(x)
    p2, l2, _ = m.forward_device(x, batch_size=bs)
    assert torch.equal(p, p2) and torch.equal(l, l2)
    # a frame alone gives the row it gives in the batch
    i = data.draw(st.integers(0, n - 1))
    p1, l1, _ = m.forward_device(x[i:i + 1].contiguous())
    assert torch.equal(p1[0], p[i]) and int(l1[0]) == int(l[i])


@settings(**_S)
@given(st.sampled_from(["deployed3-f32", "deployed10-bf16", "vtcnn2-bf16"]), st.integers(0, 5000), st.integers(1, 5000))
def test_host_driver_equals_the_device_path(key, n, chunk):
    m = _model(key)
    n = _limit(key, n)
    x = synthetic_frames(n, seed=n + 1)
    probs = np.empty((n, m.topology.classes), np.float32)
    labels = np.empty((n,), np.int32)
    _cabi.check(_cabi.lib().mdc_predict_host(m._engine(), x.ctypes.data, n, probs.ctypes.data, labels.ctypes.data, chunk))
    p, l, _ = m.forward_device(torch.from_numpy(x).cuda())
    np.testing.assert_array_equal(probs, p.cpu().numpy())
    np.testing.assert_array_equal(labels, l.cpu().numpy())


@settings(**{**_S, "max_examples": _S["max_examples"] if _N else 25})
@given(st.sampled_from(["vtcnn2-bf16", "vtcnn2-fp8"]), st.integers(2049, 9000), st.data())
def test_fused_head_batches_equal_any_chunking_across_the_small_batch_threshold(key, n, data):
    """Round 3: above 2,048 frames a 16-bit VT-CNN2 call runs dense2 + softmax + argmax inside the dense1 GEMM's epilogue;
    at or below it the small-batch dense1 forms write the hidden layer and the head is its own launch.  A batch cut into
    calls on either side of that threshold -- ragged last tiles included -- must give the same bits as one call."""
    m = _model(key)
    x = synthetic_frames(n, seed=n, device="cuda")
    p, l, _ = m.forward_device(x, batch_size=n)                       # one call: fused head
    bs = data.draw(st.sampled_from([17, 256, 1000, 2048, 2049, 2305, 4096]))
    p2, l2, _ = m.forward_device(x, batch_size=bs)
    assert torch.equal(p, p2) and torch.equal(l, l2), (n, bs)
    p3, l3, _ = m.forward_device(x, tap="dense")                      # the unfused batch path (a logit tap keeps the head separate)
    assert torch.equal(p, p3) and torch.equal(l, l3)
