#!/usr/bin/env python3
"""rocprofv3 driver: ten epochs of the training step (csrc/train.hip) at the reference's geometry -- 18,900 frames, batches of
1,024 (CNN.ipynb cell 5 / 7) -- for the two deployed nets and cnn.py's literal one, plus the validation pass of each epoch.
    rocprofv3 --kernel-trace --stats -- python3 tools/prof_train.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulationdetectioncnn_amd import Topology, synthetic_frames, synthetic_weights
from modulationdetectioncnn_amd.training import Trainer
n, nv, batch = 18900, 8100, 1024
for topo in (Topology.deployed(3), Topology.deployed(10), Topology.cnnpy(10, 10, 5)):
    x = synthetic_frames(n + nv, seed=2016, device="cuda:0") * (40.0 if topo.kind == "cnnpy" else 1.0)
    lab = torch.randint(0, topo.classes, (n + nv,), device="cuda:0")
    tr = Trainer(topo, synthetic_weights(topo, seed=2016), device=0)
    xd, yd = tr._frames(x[:n]), tr._targets(lab[:n], n)
    xv, yv = tr._frames(x[n:]), tr._targets(lab[n:], nv)
    for ep in range(10):
        order = torch.randperm(n, device="cuda:0").to(torch.int32)
        for s in range(0, n, batch):
            tr.train_batch(xd, yd, order, s, min(batch, n - s))
        tr.evaluate_enqueue(xv, yv)
        tr.read()
    tr.close()
torch.cuda.synchronize()
