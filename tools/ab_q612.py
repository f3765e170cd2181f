#!/usr/bin/env python3
"""ON THE GPU BOX: the Q6.12 integer forward (mdc_forward_q612) for two BUILDS of the library in interleaved child
processes -- frames/s at 2^20 frames and one sha of the integer outputs (they must not change).
    gpurun -- 'python tools/ab_q612.py tools/ab_prev.so [rounds = 3]'"""
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
from modulationdetectioncnn_amd import VTCNN2, synthetic_frames
G = os.path.join(%r, "tests", "golden", "weights")
x = synthetic_frames(1 << 20, seed=2016, sigma=0.3, device="cuda:0")
for net in ("3", ""):
    m = VTCNN2.from_npz(os.path.join(G, net + "convmodrecnets_CNN2_0.5.npz"), device=0)
    d, l = m.predict_q612(x, as_float=False)
    torch.cuda.synchronize()
    sha = hashlib.sha1(d.cpu().numpy().tobytes() + l.cpu().numpy().tobytes()).hexdigest()[:10]
    t0 = time.perf_counter()
    for _ in range(20): m.predict_q612(x, as_float=False)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 20
    print("RES F=%%s %%s %%.3f ms  %%.3e frames/s" %% (net or "10", sha, el * 1e3, (1 << 20) / el), flush=True)
''' % (ROOT, ROOT)
for rnd in range(rounds):
    for name, lib in (("prev   ", other), ("current", "current")):
        r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        for line in r.stdout.splitlines():
            if line.startswith("RES"):
                print(f"round {rnd} {name}", line[4:], flush=True)
        if r.returncode != 0:
            print(r.stderr[-600:]); sys.exit(1)
