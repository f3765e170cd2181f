#!/usr/bin/env python3
"""ON THE GPU BOX: A/B of two BUILDS of the library in one run (boxes differ by up to 5 % in the conv kernel's time, so a
1-2 % change can only be judged on one box): interleaved rounds, child processes, per-kernel HIP-event times of the
VT-CNN2 forward at 2^20 frames.  The second build is any other .so, typically the one saved before an edit:
    cp modulationdetectioncnn_amd/libmdc.so gpurun_out/libmdc_prev.so        # before rebuilding
    gpurun -- 'python tools/ab_libs.py tools/ab_prev.so [dtypes = bf16,fp8] [rounds = 3] [frames per call = 2^20]'
(gpurun_out/ does not travel to the box: copy the saved library to tools/ab_prev.so -- git-ignored -- first.)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other = os.path.abspath(sys.argv[1])
dtypes = sys.argv[2] if len(sys.argv) > 2 else "bf16,fp8"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 20      # frames per mdc_forward call (65536 = the library's default)
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
from modulationdetectioncnn_amd import _cabi
if sys.argv[1] != "current":
    _cabi.LIB_PATHS["product"] = sys.argv[1]
import torch, hashlib
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames
x = synthetic_frames(1 << 20, seed=2016, device="cuda:0")
G = os.path.join(%r, "tests", "golden", "weights")
for dt in sys.argv[2].split(","):
    if dt.startswith("dep"):      # dep3-f32, dep10-bf16, ...: the bundled deployed nets (per-kernel time = the whole forward)
        net, dd = dt.split("-")
        m = VTCNN2.from_npz(os.path.join(G, ("3" if net == "dep3" else "") + "convmodrecnets_CNN2_0.5.npz"), device=0, dtype=dd)
    else:
        m = VTCNN2.synthetic(Topology.vtcnn2(11), seed=2016, device=0, dtype=dt)
    CH = int(sys.argv[3])
    p, l, _ = m.forward_device(x, batch_size=CH)
    torch.cuda.synchronize()
    sha = hashlib.sha1(p.cpu().numpy().tobytes() + l.cpu().numpy().tobytes()).hexdigest()[:10]
    m.set_profiling(True)
    REPS = 40 if dt.startswith("dep") else 8
    for _ in range(REPS): m.forward_device(x, probs=p, labels=l, batch_size=CH)
    torch.cuda.synchronize()
    prof = {k: v[0] / REPS for k, v in m.read_profile().items()}      # per 2^20 frames, however many launches that took
    print("RES", dt, sha, " ".join(f"{k[7:]} {v:.3f}" for k, v in prof.items()), "sum %%.3f" %% sum(prof.values()), flush=True)
    del m
''' % (ROOT, ROOT)
for rnd in range(rounds):
    for name, lib in (("prev   ", other), ("current", "current")):
        r = subprocess.run([sys.executable, "-c", CHILD, lib, dtypes, str(chunk)], capture_output=True, text=True)
        for line in r.stdout.splitlines():
            if line.startswith("RES"):
                print(f"round {rnd} {name}", line[4:], flush=True)
        if r.returncode != 0:
            print(r.stderr[-600:]); sys.exit(1)
