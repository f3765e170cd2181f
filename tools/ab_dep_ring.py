#!/usr/bin/env python3
"""ON THE GPU BOX: the 3-filter f32 net with its frames fed by direct loads (MDC_DEP_RING=0, the round-2 kernel) and through
the per-wave asm-issued LDS-DMA ring at depths 2, 3, 4, 6, 8 (alternates build), interleaved rounds in child processes:
frames/s at 2^20 and 2^21 frames by HIP events, and the sha of probabilities + labels (must be ONE value).
usage: ab_dep_ring.py [rounds = 3] [depths = 0,2,3,4,6,8]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
DEPTHS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "2", "3", "4", "6", "8"]
CHILD = r'''
import sys, os, torch, hashlib
sys.path.insert(0, %r)
from modulationdetectioncnn_amd import VTCNN2, synthetic_frames
m = VTCNN2.from_npz(os.path.join(%r, "tests", "golden", "weights", "3convmodrecnets_CNN2_0.5.npz"), device=0, _lib_variant="alternates")
out = []
for logn in (20, 21):
    x = synthetic_frames(1 << logn, seed=2016, device="cuda:0")
    p, l, _ = m.forward_device(x)
    torch.cuda.synchronize()
    sha = hashlib.sha1(p.cpu().numpy().tobytes() + l.cpu().numpy().tobytes()).hexdigest()[:12]
    for _ in range(5): m.forward_device(x, probs=p, labels=l)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): m.forward_device(x, probs=p, labels=l)
    e1.record(); torch.cuda.synchronize()
    out.append((logn, sha, (1 << logn) * 40 / (e0.elapsed_time(e1) * 1e-3)))
    del x, p, l
print("RES", " ".join(f"2^{a}: {c:.4g} frames/s sha {b}" for a, b, c in out))
''' % (ROOT, ROOT)
shas = set()
for rnd in range(ROUNDS):
    for ring in DEPTHS:
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MDC_DEP_RING=ring), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RES")]
        print(f"round {rnd} ring {ring}:", *line, flush=True)
        if r.returncode != 0:
            print(r.stderr[-800:]); sys.exit(1)
        shas |= {w for l in line for i, w in enumerate(l.split()) if i and l.split()[i - 1] == "sha"}
print("bit-identical across ring depths:", len(shas) == 2, shas)      # one value per batch size
