#!/usr/bin/env python3
"""ON THE GPU BOX: cnn.py's literal model (mdc_dense_chain<2,3>) and the VT-CNN2 head (mdc_dense_chain<1,1>, as its own launch
through the dense tap) for two BUILDS of the library in interleaved child processes -- frames/s at 2^20 frames and one sha of
the outputs (they must not change).
    gpurun -- 'python tools/ab_cnnpy.py tools/ab_prev.so [rounds = 3]'"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
CHILD = r'''
import sys, os, hashlib, time
sys.path.insert(0, %r)
from modulationdetectioncnn_amd import _cabi
if sys.argv[1] != "current":
    _cabi.LIB_PATHS["product"] = sys.argv[1]
import torch
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames
x = synthetic_frames(1 << 20, seed=2016, device="cuda:0")
m = VTCNN2.synthetic(Topology.cnnpy(10, 10, 5), seed=2016, device=0)
p, l, _ = m.forward_device(x)
torch.cuda.synchronize()
sha = hashlib.sha1(p.cpu().numpy().tobytes() + l.cpu().numpy().tobytes()).hexdigest()[:10]
m.set_profiling(True)
for _ in range(20): m.forward_device(x, probs=p, labels=l)
torch.cuda.synchronize()
(name, (ms, cnt)), = m.read_profile().items()
print("RES cnnpy %%s %%.3f ms  %%.3e frames/s  %%.2f TB/s" %% (sha, ms / cnt, (1 << 20) / (ms / cnt * 1e-3), (1 << 20) * 1044 / (ms / cnt * 1e-3) / 1e12), flush=True)
'''  % ROOT
for rnd in range(rounds):
    for name, lib in (("prev   ", other), ("current", "current")):
        r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        for line in r.stdout.splitlines():
            if line.startswith("RES"):
                print(f"round {rnd} {name}", line[4:], flush=True)
        if r.returncode != 0:
            print(r.stderr[-600:]); sys.exit(1)
