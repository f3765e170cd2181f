#!/usr/bin/env python3
"""Small driver for rocprofv3: a few forwards of the bf16 VT-CNN2 path on 65,536 frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
m = VTCNN2.synthetic(Topology.vtcnn2(11), seed=2016, device=0, dtype=dtype)
x = synthetic_frames(n, seed=2016, device="cuda:0")
for _ in range(12):
    m.forward_device(x)
torch.cuda.synchronize()
if len(sys.argv) > 3:      # print the HIP-event kernel times too
    m.set_profiling(True)
    for _ in range(8):
        m.forward_device(x)
    torch.cuda.synchronize()
    print({k: round(v[0] / v[1], 4) for k, v in m.read_profile().items()})
