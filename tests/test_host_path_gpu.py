"""model.predict(X) on HOST buffers (cnn.py:198: X_test is a numpy array): mdc_predict_host / mdc_predict_host_iq_u8 --
pinned ring, copy / compute / result streams -- must return exactly what the device entry points return for the same
frames, whatever the chunk size, the number of chunks in flight or the kind of host memory."""
import ctypes

import numpy as np
import pytest
import torch

import modulationdetectioncnn_amd.model as model_mod
from conftest import load_deployed_npz
from modulationdetectioncnn_amd import VTCNN2, Topology, _cabi, synthetic_frames

pytestmark = pytest.mark.gpu


def _t1(dtype="f32"):
    m = VTCNN2(Topology.deployed(3, 3), dtype=dtype)
    m.set_weights(load_deployed_npz("3convmodrecnets_CNN2_0.5"))
    return m


def _device_result(m, x_np):
    p, l, _ = m.forward_device(torch.from_numpy(x_np).cuda())
    return p.cpu().numpy(), l.cpu().numpy()


@pytest.mark.parametrize("n,chunk", [(0, 0), (1, 0), (1000, 0), (1000, 1), (1000, 333), (70001, 0), (70001, 4096), (200000, 65536)])
def test_host_frames_match_the_device_path_bit_for_bit(n, chunk):
    """1 .. 245 chunks through the three slots: every slot is reused, the last chunk is ragged, n < chunk, chunk = 1."""
    m = _t1()
    x = np.ascontiguousarray(synthetic_frames(n, seed=5) * np.float32(3.0))
    probs = np.full((n, 3), np.nan, np.float32)
    labels = np.full((n,), -1, np.int32)
    _cabi.check(_cabi.lib().mdc_predict_host(m._engine(), x.ctypes.data, n, probs.ctypes.data, labels.ctypes.data, chunk))
    want_p, want_l = _device_result(m, x)
    np.testing.assert_array_equal(probs, want_p)
    np.testing.assert_array_equal(labels, want_l)


def test_keras_style_calls_take_the_host_driver(monkeypatch):
    """predict / predict_classes on numpy input: same numbers as on a device tensor, at every batch_size the reference
    passes (1024, cnn.py:176) -- which only sets a lower bound on the driver's slot length."""
    m = _t1()
    x = synthetic_frames(50000, seed=6)
    want_p, want_l = _device_result(m, x)
    np.testing.assert_array_equal(m.predict(x, batch_size=1024), want_p)
    np.testing.assert_array_equal(m.predict_classes(x, batch_size=1024), want_l)
    monkeypatch.setattr(model_mod, "HOST_MIN_CHUNK", 1)
    np.testing.assert_array_equal(m.predict(x, batch_size=999), want_p)            # 51 chunks, the last one ragged
    p, l = m.predict_host(x)
    np.testing.assert_array_equal(p, want_p)
    np.testing.assert_array_equal(l, want_l)
    assert m.predict_host(x, want_probs=False)[0] is None
    with pytest.raises(ValueError):
        m.predict(np.zeros((4, 2, 127), np.float32))


def test_pinned_host_memory_is_read_in_place():
    m = _t1()
    xt = torch.from_numpy(synthetic_frames(30000, seed=7) * np.float32(2.0)).pin_memory()
    x = xt.numpy()
    want_p, want_l = _device_result(m, x)
    for _ in range(2):                                   # second call: context reused
        p, l = m.predict_host(x, batch_size=8192)
        np.testing.assert_array_equal(p, want_p)
        np.testing.assert_array_equal(l, want_l)


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_host_frames_vtcnn2(dtype):
    """the canonical net: workspace owned by the driver, several launch chunks, small-batch forms on the ragged tail"""
    m = VTCNN2.synthetic(Topology.vtcnn2(11), dtype=dtype)
    x = synthetic_frames(9000, seed=8)
    want_p, want_l = _device_result(m, x)
    p, l = m.predict_host(x, batch_size=4096)
    np.testing.assert_array_equal(p, want_p)
    np.testing.assert_array_equal(l, want_l)
    p, l = m.predict_host(x)                              # one chunk; the context grows (workspace, slots)
    np.testing.assert_array_equal(p, want_p)
    np.testing.assert_array_equal(l, want_l)


@pytest.mark.parametrize("hop", [128, 16, 7, 300])
@pytest.mark.parametrize("kind", ["deployed3", "deployed10-f16", "vtcnn2-bf16"])
def test_host_iq_bytes_match_the_device_path(kind, hop):
    """raw capture in host memory: windows overlap (hop < 128), abut (128) or leave gaps (300); chunk boundaries re-read
    the overlapping bytes"""
    if kind == "vtcnn2-bf16":
        m = VTCNN2.synthetic(Topology.vtcnn2(11), dtype="bf16")
        n = 3000
    else:
        m = VTCNN2.synthetic(kind.split("-")[0], dtype=kind.split("-")[1] if "-" in kind else "f32")
        n = 40000
    rng = np.random.default_rng(hop)
    iq = rng.integers(0, 256, size=2 * hop * (n - 1) + 256, dtype=np.uint8)
    scale = 0.02 / 127.5
    want_p, want_l = m.predict_iq_u8(torch.from_numpy(iq).cuda(), scale=scale, hop=hop)
    for bs in (0, 1111):
        monkey_chunk = bs
        probs, labels = np.empty((n, m.topology.classes), np.float32), np.empty((n,), np.int32)
        _cabi.check(_cabi.lib().mdc_predict_host_iq_u8(m._engine(), iq.ctypes.data, n, hop, scale, probs.ctypes.data, labels.ctypes.data, monkey_chunk))
        np.testing.assert_array_equal(probs, want_p.cpu().numpy())
        np.testing.assert_array_equal(labels, want_l.cpu().numpy())
    if hop in (128, 16):
        p, l = m.predict_iq_u8(iq, scale=scale, hop=hop)           # the numpy route of the mirror
        np.testing.assert_array_equal(p, want_p.cpu().numpy())
        np.testing.assert_array_equal(l, want_l.cpu().numpy())


@pytest.mark.parametrize("kind,hop,n", [("deployed3", 1 << 20, 128), ("deployed3", 1 << 18, 700), ("vtcnn2-bf16", 1 << 18, 520)])
def test_host_iq_bytes_with_a_hop_so_large_that_a_slot_holds_fewer_than_256_windows(kind, hop, n):
    """the library's own chunking cuts a slot to the windows that fit 64 MiB of capture (32 at hop 2^20,
    128 at 2^18); with n >= 4 slots the fill ramp is on and its 256-window floor must not exceed the slot.  Results equal
    the device path's bit for bit; the first window, the slot edges and the tail are all inside the compared range."""
    m = VTCNN2.synthetic(Topology.vtcnn2(11), dtype="bf16") if kind == "vtcnn2-bf16" else _t1()
    rng = np.random.default_rng(hop + n)
    iq = np.zeros(2 * hop * (n - 1) + 256, dtype=np.uint8) + np.uint8(127)
    for w in range(n):                                                     # only the bytes the windows read need entropy
        iq[2 * hop * w: 2 * hop * w + 256] = rng.integers(0, 256, size=256, dtype=np.uint8)
    scale = 0.02 / 127.5
    want_p, want_l = m.predict_iq_u8(torch.from_numpy(iq).cuda(), scale=scale, hop=hop)
    probs, labels = np.full((n, m.topology.classes), np.nan, np.float32), np.full((n,), -1, np.int32)
    _cabi.check(_cabi.lib().mdc_predict_host_iq_u8(m._engine(), iq.ctypes.data, n, hop, scale, probs.ctypes.data, labels.ctypes.data, 0))
    np.testing.assert_array_equal(probs, want_p.cpu().numpy())
    np.testing.assert_array_equal(labels, want_l.cpu().numpy())


def test_host_path_errors():
    m = _t1()
    L = _cabi.lib()
    L.mdc_last_error.restype = ctypes.c_char_p
    out = np.empty((4, 3), np.float32)
    assert L.mdc_predict_host(m._engine(), None, 4, out.ctypes.data, None, 0) == -22 and b"null input" in L.mdc_last_error()
    assert L.mdc_predict_host(m._engine(), out.ctypes.data, -1, None, None, 0) == -22
    assert L.mdc_predict_host(m._engine(), out.ctypes.data, 4, None, None, -5) == -22 and b"chunk" in L.mdc_last_error()
    assert L.mdc_predict_host_iq_u8(m._engine(), out.ctypes.data, 4, 0, 1.0, None, None, 0) == -22 and b"hop" in L.mdc_last_error()
    assert L.mdc_predict_host(m._engine(), None, 0, None, None, 0) == 0
    # a DEVICE pointer must not reach the staging threads' memcpy
    xd = synthetic_frames(64, seed=1, device="cuda")
    assert L.mdc_predict_host(m._engine(), xd.data_ptr(), 64, out.ctypes.data, None, 0) == -22 and b"device memory" in L.mdc_last_error()
    iqd = torch.zeros(4 * 256, dtype=torch.uint8, device="cuda")
    assert L.mdc_predict_host_iq_u8(m._engine(), iqd.data_ptr(), 4, 128, 1.0, out.ctypes.data, None, 0) == -22 and b"device memory" in L.mdc_last_error()
    cn = VTCNN2.synthetic("cnnpy")
    assert L.mdc_predict_host_iq_u8(cn._engine(), out.ctypes.data, 1, 128, 1.0, None, None, 0) == -95
    x = synthetic_frames(100, seed=1)
    np.testing.assert_array_equal(cn.predict(x), cn.predict(torch.from_numpy(x).cuda()).cpu().numpy())      # T4 through the driver too


def test_out_arrays_and_the_multi_gpu_host_driver_on_one_gpu():
    """results straight into slices of the caller's arrays; MultiStreamPredictor.predict_host with G = 1 device"""
    from modulationdetectioncnn_amd.sharding import MultiStreamPredictor
    m = _t1()
    x = synthetic_frames(20000, seed=9)
    want_p, want_l = _device_result(m, x)
    P, Lb = np.zeros((20000, 3), np.float32), np.zeros((20000,), np.int32)
    m.predict_host(x[5000:12000], out=(P[5000:12000], Lb[5000:12000]))
    np.testing.assert_array_equal(P[5000:12000], want_p[5000:12000])
    np.testing.assert_array_equal(Lb[5000:12000], want_l[5000:12000])
    assert not P[:5000].any() and not P[12000:].any()
    with pytest.raises(ValueError):
        m.predict_host(x[:10], out=(P[:10, :2], Lb[:10]))          # not contiguous / wrong shape
    msp = MultiStreamPredictor.for_models([m], streams_per_device=2)
    p, l = msp.predict_host(x)
    np.testing.assert_array_equal(p, want_p)
    np.testing.assert_array_equal(l, want_l)


def test_host_batch_beyond_4_gib_with_the_fill_ramp():
    """A host array of more than 2^32 bytes (2^22 + 3 frames) through the streaming driver with the library's own chunking:
    the fill ramp (first chunks a quarter and a half slot long), 65 full slots and a ragged tail; byte offsets into the
    caller's array need 64 bits.  Slices at the start, around the ramp's chunk edges, across the 4 GiB boundary and at the
    end must equal the device path's results for those frames."""
    n = (1 << 22) + 3
    block = synthetic_frames(1 << 16, seed=11)                       # 64 MiB, repeated with a per-repeat scale: no 4 GiB of RNG on the host
    x = np.empty((n, 2, 128), np.float32)
    for i, s in enumerate(range(0, n, 1 << 16)):
        e = min(n, s + (1 << 16))
        np.multiply(block[: e - s], np.float32(1.0 + 0.03125 * (i % 7)), out=x[s:e])
    m = _t1()
    p, l = m.predict_host(x)
    assert p.shape == (n, 3) and l.shape == (n,)
    for lo, hi in ((0, 500), (16384 - 100, 16384 + 100), (49152 - 100, 49152 + 100), ((1 << 22) - 200, n), (2097152 - 50, 2097152 + 50)):
        want_p, want_l = _device_result(m, x[lo:hi])
        np.testing.assert_array_equal(p[lo:hi], want_p)
        np.testing.assert_array_equal(l[lo:hi], want_l)
