#!/usr/bin/env python3
"""Debug aid: compare the bf16 conv features (flat tap) with the f32 kernel's and list where they disagree."""
import os, sys, collections, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames, synthetic_weights
topo = Topology.vtcnn2(3); w = synthetic_weights(topo, seed=2016)
mb = VTCNN2(topo, device=0, dtype="bf16"); mb.set_weights(w)
mf = VTCNN2(topo, device=0, dtype="f32"); mf.set_weights(w)
for n in [int(a) for a in sys.argv[1:]] or [64, 4096]:
    x = synthetic_frames(n, seed=7, device="cuda:0")
    ref = mf.predict(x, tap="flat")
    for rep in range(2):
        flat = mb.predict(x, tap="flat")
        err = (flat - ref).abs()
        sc = float(ref.abs().max())
        bad = (err > 0.02 * sc).nonzero().cpu().numpy()
        cnt = collections.Counter((int(c % 132), int(c // 132)) for r, c in bad)
        rows = sorted(set(int(r) for r, c in bad))
        print(f"n={n} rep={rep} bad entries {len(bad)} rows {len(rows)} first rows {rows[:12]} (w,o) counts {sorted(cnt.items())[:12]}")
    if len(bad):
        flat_c = flat.cpu().numpy(); ref_c = ref.cpu().numpy()
        for r, c in bad[:6]:
            o, wpos = c // 132, c % 132
            print(f"   row {r} o {o} w {wpos}: gpu {flat_c[r, c]:.6e} ref {ref_c[r, c]:.6e}  ref(w-1) {ref_c[r, c - 1]:.6e} ref(w+1) {ref_c[r, c + 1]:.6e} gpu(w-1) {flat_c[r, c-1]:.6e} gpu(w+1) {flat_c[r, c+1]:.6e}")
