#!/usr/bin/env python3
"""Determinism soak for the kernels whose correctness rests on hand-written synchronisation (counted waits, raw barriers,
LDS-DMA rings, asm-sequenced MFMA hazards): the same forward `reps` times, every result compared bit for bit with the
first one.  A race or an unpadded hazard shows up as an occasional differing result (round 1's `acc = bias` hazard did:
"wrong output 1, now and then").  `run(...)` is also called by tests/test_soak_gpu.py with a small `reps`.
    python tools/soak.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = (  # (topology, dtype, frame counts): every small-batch form, the batch forms, the deployed LDS-DMA rings
    # (1 << 18: every CU busy with full tiles, the LDS-DMA streams at their HBM rate -- where a hole in the phased dense1's
    #  counted waits would show; fp8: the E4M3-feature form with its own wait counts)
    ("vtcnn2", "bf16", (1, 16, 17, 64, 256, 1024, 1025, 2048, 5000, 1 << 18)),
    ("vtcnn2", "fp8", (1, 64, 256, 1024, 5000, 1 << 18)),
    ("vtcnn2", "f32", (1, 128, 129, 2048)),
    ("deployed10", "bf16", (1, 1000, 70001)),
    ("deployed10", "f16", (70001,)),
    ("deployed3", "fp8", (70001,)),
    ("deployed3", "f32", (1, 70001)),
)


def run(reps=200, cases=CASES, device=0, log=print):
    import torch
    from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames
    bad = []
    for topo, dtype, sizes in cases:
        m = VTCNN2.synthetic(Topology.vtcnn2(11) if topo == "vtcnn2" else topo, device=device, dtype=dtype)
        for n in sizes:
            x = synthetic_frames(n, seed=n, device=f"cuda:{device}")
            p0, l0, _ = m.forward_device(x)
            p0, l0 = p0.clone(), l0.clone()
            h0 = m.predict(x, tap="hidden").clone() if topo == "vtcnn2" else None
            diffs = 0
            for r in range(reps):
                p, l, _ = m.forward_device(x)
                if not (torch.equal(p, p0) and torch.equal(l, l0)):
                    diffs += 1
                if h0 is not None and r % 8 == 0 and not torch.equal(m.predict(x, tap="hidden"), h0):
                    diffs += 1
            log(f"{topo} {dtype} n={n}: {reps} repeats, {diffs} differing", flush=True)
            if diffs:
                bad.append((topo, dtype, n, diffs))
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
    print("SOAK", "FAILED: " + repr(bad) if bad else "OK")
    sys.exit(1 if bad else 0)
