#!/usr/bin/env python3
"""rocprofv3 driver: a few forwards of the deployed F=3 / F=10 nets on 2^20 frames, f32, bf16 and f16 kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulationdetectioncnn_amd import VTCNN2, synthetic_frames
g = os.path.join(ROOT, "tests", "golden", "weights")
x = synthetic_frames(1 << 20, seed=2016, device="cuda:0")
for f in ("3convmodrecnets_CNN2_0.5.npz", "convmodrecnets_CNN2_0.5.npz"):
    for dt in ("f32", "bf16", "f16"):
        m = VTCNN2.from_npz(os.path.join(g, f), device=0, dtype=dt)
        for _ in range(3):
            m.forward_device(x)
torch.cuda.synchronize()
