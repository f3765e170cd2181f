#!/usr/bin/env python3
"""Debug aid: fp8-mode conv features and logits against the f32 kernels."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames, synthetic_weights
topo = Topology.vtcnn2(11); w = synthetic_weights(topo, seed=2016)
m8 = VTCNN2(topo, device=0, dtype="fp8"); m8.set_weights(w)
mb = VTCNN2(topo, device=0, dtype="bf16"); mb.set_weights(w)
mf = VTCNN2(topo, device=0, dtype="f32"); mf.set_weights(w)
for n in [int(a) for a in sys.argv[1:]] or [16, 4096]:
    x = synthetic_frames(n, seed=7, device="cuda:0")
    ref = mf.predict(x, tap="flat"); sc = float(ref.abs().max())
    for name, m in (("bf16", mb), ("fp8", m8)):
        f = m.predict(x, tap="flat")
        err = (f - ref).abs()
        lg = m.predict(x, tap="dense"); lr = mf.predict(x, tap="dense")
        lab = m.predict_classes(x); labr = mf.predict_classes(x)
        print(f"n={n} {name}: feature max err {float(err.max())/sc:.4f} of max, rms {float((err**2).mean().sqrt())/sc:.5f}; logits max err "
              f"{float((lg-lr).abs().max())/float(lr.abs().max()):.4f} of max; label agreement {float((lab==labr).float().mean()):.4f}; nan {int(torch.isnan(f).sum())}")
