#!/usr/bin/env python3
"""ON THE GPU BOX: per-kernel HIP-event times of the VT-CNN2 forward (2^20 frames, one launch of each kernel) per dtype.
usage: time_vt.py [dtypes = bf16,fp8] [reps = 6]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames
dtypes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16", "fp8"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
x = synthetic_frames(1 << 20, seed=2016, device="cuda:0")
for dt in dtypes:
    m = VTCNN2.synthetic(Topology.vtcnn2(11), seed=2016, device=0, dtype=dt)
    n = 1 << (16 if dt == "f32" else 20)
    xs = x[:n]
    for _ in range(2): m.forward_device(xs, batch_size=n)
    m.set_profiling(True)
    for _ in range(reps): m.forward_device(xs, batch_size=n)
    torch.cuda.synchronize()
    prof = {k: round(v[0] / max(v[1], 1), 4) for k, v in m.read_profile().items()}
    tot = sum(prof.values())
    print(dt, prof, "sum %.3f ms -> %.4g frames/s" % (tot, n / tot * 1e3), flush=True)
    del m
