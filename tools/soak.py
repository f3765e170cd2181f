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
    # cnn.py's literal net (dense_chain<2,3>): four k-blocks per tile refilled progressively behind counted vmcnt waits (round 5);
    # one tile per wave, ragged last tiles, and enough tiles that every wave walks several with a next tile in flight
    ("cnnpy", "f32", (1, 16, 17, 1000, 70001, 1 << 18, 1 << 20)),
)


def run(reps=200, cases=CASES, device=0, log=print):
    import torch
    from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames
    bad = []
    for topo, dtype, sizes in cases:
        m = VTCNN2.synthetic(Topology.vtcnn2(11) if topo == "vtcnn2" else Topology.cnnpy(10, 10, 5) if topo == "cnnpy" else topo, device=device, dtype=dtype)
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


def run_training(reps=50, device=0, log=print):
    """The training step (csrc/train.hip): the same epoch of 19 mini-batches from the same state `reps` times, eagerly and as
    a replayed hipGraph, with and without the optional Dropout -- weights, Adam moments and the step count compared bit for bit
    with the first run.  What it would catch: a hole in the reduce + Adam launch's ticket (the step count bumped before every
    work-group has read it), a partial vector read before its wave wrote it, a mask keyed on a stale step."""
    import torch
    from modulationdetectioncnn_amd import Topology, synthetic_frames, synthetic_weights
    from modulationdetectioncnn_amd.training import Trainer
    bad = []
    n, batch = 18900, 1024
    for name, topo in (("deployed3", Topology.deployed(3)), ("deployed10", Topology.deployed(10)), ("cnnpy", Topology.cnnpy(10, 10, 5))):
        x = synthetic_frames(n, seed=5, device=f"cuda:{device}") * (40.0 if topo.kind == "cnnpy" else 1.0)
        lab = torch.randint(0, topo.classes, (n,), device=f"cuda:{device}", generator=torch.Generator(f"cuda:{device}").manual_seed(3))
        order = torch.randperm(n, device=f"cuda:{device}", generator=torch.Generator(f"cuda:{device}").manual_seed(4)).to(torch.int32)
        w0 = synthetic_weights(topo, seed=9, bias_scale=0.02)
        for dropout in (0.0, 0.5):
            tr = Trainer(topo, w0, device=device, dropout=dropout, dropout_seed=11)
            xd, yd = tr._frames(x), tr._targets(lab, n)
            zeros = [(k * 0, b * 0) for k, b in w0]

            def reset():
                tr.set_weights(w0)
                tr.set_optimizer_state({"iterations": 0, "m": zeros, "v": zeros})

            def epoch():
                for s in range(0, n, batch):
                    tr.train_batch(xd, yd, order, s, min(batch, n - s))

            def state():
                st = tr.optimizer_state()
                return [t for pair in tr.get_weights() + st["m"] + st["v"] for t in pair], st["iterations"]

            reset(); epoch(); torch.cuda.synchronize()
            ref, it_ref = state()
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                tr.read()
                with torch.cuda.graph(g, stream=side):
                    epoch()
            torch.cuda.current_stream(device).wait_stream(side)
            diffs = 0
            for r in range(reps):
                reset()
                torch.cuda.synchronize()
                if r % 2:
                    g.replay()
                else:
                    epoch()
                torch.cuda.synchronize()
                got, it = state()
                if it != it_ref or any((a != b).any() for a, b in zip(got, ref)):
                    diffs += 1
            log(f"train {name} dropout={dropout}: {reps} epochs of 19 steps (eager and graph replay alternating), {diffs} differing", flush=True)
            if diffs:
                bad.append(("train", name, dropout, diffs))
            tr.close()
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
    bad += run_training(max(10, (int(sys.argv[1]) if len(sys.argv) > 1 else 2000) // 10))
    print("SOAK", "FAILED: " + repr(bad) if bad else "OK")
    sys.exit(1 if bad else 0)
