import sys, time, torch
sys.path.insert(0, '/root/repo')
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames
m = VTCNN2.synthetic(Topology.vtcnn2(11), seed=2016, device=0, dtype="bf16")
n = 1 << 20
x = synthetic_frames(n, seed=2016, device="cuda:0")
probs = torch.empty((n, 11), dtype=torch.float32, device=x.device); labels = torch.empty((n,), dtype=torch.int32, device=x.device)
for chunk in (32768, 65536, 131072, 262144, 65536):
    for _ in range(2): m.forward_device(x, probs, labels, batch_size=chunk)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): m.forward_device(x, probs, labels, batch_size=chunk)
    torch.cuda.synchronize(); el = (time.perf_counter() - t) / 5
    print(f"chunk {chunk}: {el*1e3:.2f} ms per 2^20 frames -> {n/el:.4g} frames/s", flush=True)
