#!/usr/bin/env python3
"""A/B of the dense1 kernels in separate processes, interleaved rounds: the product library (phased GEMM with the head
fused into its epilogue) against the alternates test build (libmdc_alt.so) with the head as its own launch
(MDC_D1_FUSED_HEAD=0) and with the one-barrier-per-K-tile GEMM (MDC_DENSE1_PHASED=0).  Bit-equality of the hidden layer
AND of the probabilities / labels on a full-size batch over several repeats -- a race in the phased kernel's LDS-DMA
ordering or in the fused epilogue's LDS reuse would show as a hash that comes and goes -- and the kernels' times.
usage: ab_dense1.py [log2 frames = 18] [rounds = 2] [dtype = bf16]
dtype f32 (round 5): the 128 x 256-tile f32 kernel with its fused head against rounds 1-4's 128 x 128 tiles + head launch
(MDC_DENSE1_PHASED=0 selects those in the alternates build too) and against its own unfused form."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOGN = int(sys.argv[1]) if len(sys.argv) > 1 else 18
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
DTYPE = sys.argv[3] if len(sys.argv) > 3 else "bf16"
CHILD = r'''
import sys, torch, hashlib
sys.path.insert(0, %r)
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames
variant = sys.argv[1]
m = VTCNN2.synthetic(Topology.vtcnn2(11), seed=2016, device=0, dtype=%r, _lib_variant=variant)
x = synthetic_frames(1 << %d, seed=2016, device="cuda:0")
sha = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()
for rep in range(2):
    h = m.predict(x, tap="hidden", batch_size=1 << 20)
    p, l, _ = m.forward_device(x, batch_size=1 << 20)
    torch.cuda.synchronize()
    print("HASH", rep, sha(h), sha(p), sha(l), flush=True)
del h
m.set_profiling(True)
for _ in range(4): m.forward_device(x, batch_size=1 << 20)
torch.cuda.synchronize()
print("PROF", {k: round(v[0] / max(v[1], 1), 4) for k, v in m.read_profile().items()})
''' % (ROOT, DTYPE, LOGN)
VARIANTS = {"product (fused head)": ("product", {}),
            "alternates: head as its own launch": ("alternates", {"MDC_D1_FUSED_HEAD": "0"}),
            "alternates: one-barrier dense1": ("alternates", {"MDC_DENSE1_PHASED": "0"})}
hashes = set()
for rnd in range(ROUNDS):
    for name, (variant, env) in VARIANTS.items():
        r = subprocess.run([sys.executable, "-c", CHILD, variant], env=dict(os.environ, **env), capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith(("HASH", "PROF"))]
        print(f"round {rnd} {name}:", *lines, sep="\n  ", flush=True)
        if r.returncode != 0:
            print(r.stderr[-800:]); sys.exit(1)
        hashes |= {tuple(l.split()[2:]) for l in lines if l.startswith("HASH")}
ok = len(hashes) == 1
print("bit-identical across kernels and repeats:", ok)
sys.exit(0 if ok else 1)
