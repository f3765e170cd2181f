#!/usr/bin/env python3
"""Small-batch latency (the reference's deployment classifies one window at a time): wall time of one forward call,
enqueue to result-on-device, for n = 1 .. 4096 frames already resident in HBM; median of 200 calls."""
import os, sys, time, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames

for name, make in (("T1 f32", lambda: VTCNN2.synthetic("deployed3", device=0)),
                   ("T2 f32", lambda: VTCNN2.synthetic("deployed10", device=0)),
                   ("T2 bf16", lambda: VTCNN2.synthetic("deployed10", device=0, dtype="bf16")),
                   ("T3 bf16", lambda: VTCNN2.synthetic(Topology.vtcnn2(11), device=0, dtype="bf16")),
                   ("T3 f32", lambda: VTCNN2.synthetic(Topology.vtcnn2(11), device=0, dtype="f32"))):
    m = make()
    row = []
    for n in (1, 16, 256, 4096):
        x = synthetic_frames(n, seed=1, device="cuda:0")
        probs = torch.empty((n, m.topology.classes), dtype=torch.float32, device="cuda"); labels = torch.empty((n,), dtype=torch.int32, device="cuda")
        for _ in range(20): m.forward_device(x, probs, labels)
        torch.cuda.synchronize()
        ts = []
        for _ in range(200):
            t = time.perf_counter(); m.forward_device(x, probs, labels); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        row.append(f"n={n}: {statistics.median(ts)*1e6:.0f} us")
    print(f"{name}: " + ", ".join(row), flush=True)
