#!/usr/bin/env python3
"""ON THE GPU BOX: A/B of two BUILDS of the library on the training step (csrc/train.hip) in one run: interleaved child
processes; per net (T1, T2, T4) the time of one 1,024-frame step (gradient + Adam launches) replayed from a hipGraph of 19
steps (an epoch of the reference's 18,900 frames), the eager time, and a hash of the weights after those steps (a change
that only moves loads and stores must leave it unchanged).
    cp modulationdetectioncnn_amd/libmdc.so tools/ab_prev.so        # before rebuilding
    gpurun -- 'python tools/ab_train.py tools/ab_prev.so [rounds = 3]'"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
CHILD = r'''
import sys, os, hashlib
sys.path.insert(0, %r)
from modulationdetectioncnn_amd import _cabi
if sys.argv[1] != "current":
    _cabi.LIB_PATHS["product"] = sys.argv[1]
import torch, numpy as np
from modulationdetectioncnn_amd import Topology, synthetic_frames, synthetic_weights
from modulationdetectioncnn_amd.training import Trainer
n, batch = 18900, 1024
for name, topo in (("T1", Topology.deployed(3)), ("T2", Topology.deployed(10)), ("T4", Topology.cnnpy(10, 10, 5))):
    x = synthetic_frames(n, seed=2016, device="cuda:0") * (40.0 if topo.kind == "cnnpy" else 1.0)
    lab = torch.randint(0, topo.classes, (n,), device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(1))
    tr = Trainer(topo, synthetic_weights(topo, seed=2016), device=0)
    xd, yd = tr._frames(x), tr._targets(lab, n)
    order = torch.randperm(n, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(2)).to(torch.int32)
    def epoch():
        for s in range(0, n, batch):
            tr.train_batch(xd, yd, order, s, min(batch, n - s))
    epoch(); tr.read(); torch.cuda.synchronize()
    sha = hashlib.sha1(b"".join(k.tobytes() + b.tobytes() for k, b in tr.get_weights())).hexdigest()[:10]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20): epoch()
    ev[1].record(); torch.cuda.synchronize()
    eager = ev[0].elapsed_time(ev[1]) / (20 * 19) * 1e3
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        tr.read()
        with torch.cuda.graph(g, stream=side):
            epoch()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    times = []
    for _ in range(30):
        ev[0].record(); g.replay(); ev[1].record(); torch.cuda.synchronize()
        times.append(ev[0].elapsed_time(ev[1]) / 19 * 1e3)
    print("RES", name, sha, "graph %%.2f us/step (min %%.2f)  eager %%.2f us/step" %% (float(np.median(times)), min(times), eager), flush=True)
    tr.close()
''' % ROOT
for rnd in range(rounds):
    for name, lib in (("prev   ", other), ("current", "current")):
        r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        for line in r.stdout.splitlines():
            if line.startswith("RES"):
                print(f"round {rnd} {name}", line[4:], flush=True)
        if r.returncode != 0:
            print(r.stderr[-800:]); sys.exit(1)
