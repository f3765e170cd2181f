#!/usr/bin/env python3
"""A/B of the two dense1 kernels (MDC_DENSE1_PHASED=0/1) in separate processes: bit-equality of the hidden layer on a
full-size batch over several repeats (a race in the phased kernel's LDS-DMA ordering would show as a mismatch that
comes and goes) and the kernel times."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch, hashlib
sys.path.insert(0, %r)
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames
m = VTCNN2.synthetic(Topology.vtcnn2(11), seed=2016, device=0, dtype="bf16")
x = synthetic_frames(1 << 18, seed=2016, device="cuda:0")
for rep in range(4):
    h = m.predict(x, tap="hidden")
    torch.cuda.synchronize()
    print("HASH", rep, hashlib.sha1(h.cpu().numpy().tobytes()).hexdigest(), flush=True)
m.set_profiling(True)
for _ in range(3): m.forward_device(x)
torch.cuda.synchronize()
print("PROF", {k: round(v[0] / v[1], 4) for k, v in m.read_profile().items()})
''' % ROOT
out = {}
for mode in ("0", "1"):
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MDC_DENSE1_PHASED=mode), capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("HASH", "PROF"))]
    print("phased =", mode, *lines, sep="\n  ")
    if r.returncode != 0:
        print(r.stderr[-800:]); sys.exit(1)
    out[mode] = [l.split()[2] for l in lines if l.startswith("HASH")]
ok = len(set(out["0"] + out["1"])) == 1
print("bit-identical across kernels and repeats:", ok)
sys.exit(0 if ok else 1)
