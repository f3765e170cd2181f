"""The forward entry points enqueue onto the caller's stream and nothing else -- no synchronisation, no allocation, no
host read-back -- so a serving stack may capture them into a hipGraph and replay it (the reference's deployment runs one
window per start pulse through fixed buffers, cnn_test_latest1.sv:144-209: a fixed-buffer replay is that loop).  Captured
through torch.cuda.CUDAGraph (hipStreamBeginCapture on ROCm); the replay must be bit-identical to a direct call on the
same inputs, for every kernel family and for the small-batch and the batch launch forms."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames      # noqa: E402

CASES = [("deployed3", "f32"), ("deployed10", "f32"), ("deployed3", "bf16"), ("deployed10", "fp8"), ("deployed10", "f16"),
         ("vtcnn2", "f32"), ("vtcnn2", "bf16"), ("vtcnn2", "fp8"), ("cnnpy", "f32")]


def _model(topo, dtype):
    t = Topology.vtcnn2(11) if topo == "vtcnn2" else topo
    return VTCNN2.synthetic(t, seed=2016, device=0, dtype=dtype)


@pytest.mark.parametrize("topo,dtype", CASES)
@pytest.mark.parametrize("n", [1, 300, 4096])
def test_forward_is_capturable_and_replays_bit_identically(topo, dtype, n):
    m = _model(topo, dtype)
    dev = torch.device("cuda:0")
    x = synthetic_frames(n, seed=5, device="cuda:0")
    probs = torch.empty((n, m.topology.classes), dtype=torch.float32, device=dev)
    labels = torch.empty((n,), dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        m.forward_device(x, probs, labels)      # warm: workspace allocated, attributes set before the capture starts
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        m.forward_device(x, probs, labels)
    for seed in (6, 7):
        x.copy_(synthetic_frames(n, seed=seed, device="cuda:0"))      # same buffers, new windows
        probs.zero_()
        labels.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        p_graph, l_graph = probs.clone(), labels.clone()
        p_direct, l_direct, _ = m.forward_device(x)
        torch.cuda.synchronize()
        assert torch.equal(p_graph, p_direct) and torch.equal(l_graph, l_direct), (topo, dtype, n, seed)
        assert int(l_graph.min()) >= 0


def test_raw_iq_forward_is_capturable():
    m = _model("deployed3", "f32")
    iq = torch.from_numpy(np.random.default_rng(1).integers(0, 256, size=2 * (128 + 16 * 499), dtype=np.uint8)).cuda()
    p0, l0 = m.predict_iq_u8(iq, 0.02 / 127.5, hop=16)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        p1, l1 = m.predict_iq_u8(iq, 0.02 / 127.5, hop=16)
    p1.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(p0, p1) and torch.equal(l0, l1)


def test_a_captured_workspace_survives_eviction_and_growth():
    """the graph holds the scratch buffer's ADDRESS.  Five other streams forwarding through the same model push
    the capture stream's entry out of the per-stream LRU (MAX_WORKSPACES = 4), and a later, larger chunk replaces it;
    the captured buffer must stay allocated (nobody else may be handed its memory) and the replay stay bit-identical."""
    m = _model("vtcnn2", "bf16")
    dev = torch.device("cuda:0")
    n = 4096
    x = synthetic_frames(n, seed=5, device="cuda:0")
    probs = torch.empty((n, 11), dtype=torch.float32, device=dev)
    labels = torch.empty((n,), dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        m.forward_device(x, probs, labels)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        m.forward_device(x, probs, labels)
    assert len(m._ws_captured) == 1
    held = m._ws_captured[0]
    addr = held.data_ptr()
    others = [torch.cuda.Stream(device=dev) for _ in range(5)]
    for s in others:                                                  # evicts the capture stream's LRU entry
        with torch.cuda.stream(s):
            m.forward_device(x)
    with torch.cuda.stream(side):                                     # a larger chunk on the capture stream replaces the entry
        m.forward_device(synthetic_frames(3 * n, seed=9, device="cuda:0"), batch_size=3 * n)
    torch.cuda.synchronize()
    assert side.cuda_stream not in m._ws or m._ws[side.cuda_stream] is not held
    assert m._ws_captured[0] is held and held.data_ptr() == addr      # still allocated, same address
    # whatever the allocator hands out now must not alias the captured buffer
    grab = [torch.full((held.numel(),), 0xAB, dtype=torch.uint8, device=dev) for _ in range(3)]
    assert all(t.data_ptr() + t.numel() <= addr or t.data_ptr() >= addr + held.numel() for t in grab)
    x.copy_(synthetic_frames(n, seed=6, device="cuda:0"))
    probs.zero_()
    g.replay()
    torch.cuda.synchronize()
    p_direct, l_direct, _ = m.forward_device(x)
    torch.cuda.synchronize()
    assert torch.equal(probs, p_direct) and torch.equal(labels, l_direct)
    assert all(bool((t == 0xAB).all()) for t in grab)                 # the replay wrote nothing outside its own buffer
    del g
    m.release_captured_workspaces()
    assert m._ws_captured == []
