#!/usr/bin/env python3
"""Small-batch latency (the reference's deployment classifies ONE window per start pulse, cnn_test_latest1.sv:144-209):
wall time of one forward call, enqueue to result-on-device, for n = 1 .. 4096 frames already resident in HBM; median of
`reps` calls.  `rows()` is what bench.py reports as its "latency" extra leg.  `graph=True` adds the same call captured
once into a hipGraph (torch.cuda.CUDAGraph) and replayed: what a serving loop with fixed buffers pays per window."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODELS = (("T1 f32", "deployed3", "f32"), ("T2 f32", "deployed10", "f32"), ("T2 bf16", "deployed10", "bf16"),
          ("T3 f32", "vtcnn2", "f32"), ("T3 bf16", "vtcnn2", "bf16"), ("T3 fp8", "vtcnn2", "fp8"))


def rows(device=0, sizes=(1, 16, 64, 256, 4096), reps=200, models=MODELS, graph=False):
    import torch
    from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames
    out = []
    for name, topo, dtype in models:
        m = VTCNN2.synthetic(Topology.vtcnn2(11) if topo == "vtcnn2" else topo, device=device, dtype=dtype)
        us, gus = {}, {}
        for n in sizes:
            x = synthetic_frames(n, seed=1, device=f"cuda:{device}")
            probs = torch.empty((n, m.topology.classes), dtype=torch.float32, device=x.device)
            labels = torch.empty((n,), dtype=torch.int32, device=x.device)
            for _ in range(20):
                m.forward_device(x, probs, labels)
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                t = time.perf_counter()
                m.forward_device(x, probs, labels)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t)
            us[str(n)] = round(statistics.median(ts) * 1e6, 1)
            if graph:
                side = torch.cuda.Stream(device=x.device)
                with torch.cuda.stream(side):
                    m.forward_device(x, probs, labels)
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    m.forward_device(x, probs, labels)
                for _ in range(20):
                    g.replay()
                torch.cuda.synchronize()
                ts = []
                for _ in range(reps):
                    t = time.perf_counter()
                    g.replay()
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t)
                gus[str(n)] = round(statistics.median(ts) * 1e6, 1)
        row = {"model": name, "median_us_by_frames": us}
        if graph:
            row["graph_replay_median_us_by_frames"] = gus
        out.append(row)
    return out


if __name__ == "__main__":
    for r in rows(graph=True):
        print(f"{r['model']}: " + ", ".join(f"n={k}: {v:.0f} us" for k, v in r["median_us_by_frames"].items()), flush=True)
        print(f"{r['model']} (hipGraph replay): " + ", ".join(f"n={k}: {v:.0f} us" for k, v in r["graph_replay_median_us_by_frames"].items()), flush=True)
