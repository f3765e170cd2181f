#!/usr/bin/env python3
"""Raw SDR bytes -> labels on the deployed nets: fused kernel (mdc_forward_iq_u8) against the two-pass path
(mdc_iq_u8_to_frames + mdc_forward) and against the f32-frame forward alone."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modulationdetectioncnn_amd import VTCNN2, frames_from_iq_u8

n = 1 << 22
iq = torch.randint(0, 256, (n * 256,), dtype=torch.uint8, device="cuda")
for topo in ("deployed3", "deployed10"):
    m = VTCNN2.synthetic(topo, seed=2016, device=0)
    x = frames_from_iq_u8(iq, 0.02 / 127.5)
    probs = torch.empty((n, 3), dtype=torch.float32, device="cuda"); labels = torch.empty((n,), dtype=torch.int32, device="cuda")

    def timed(fn, reps=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps

    t_f32 = timed(lambda: m.forward_device(x, probs, labels))
    t_two = timed(lambda: m.forward_device(frames_from_iq_u8(iq, 0.02 / 127.5), probs, labels))
    t_fused = timed(lambda: m.predict_iq_u8(iq, 0.02 / 127.5))
    for dt in ("bf16", "f16"):
        mb = VTCNN2.synthetic(topo, seed=2016, device=0, dtype=dt)
        t_fr = timed(lambda: mb.forward_device(x, probs, labels))
        t_by = timed(lambda: mb.predict_iq_u8(iq, 0.02 / 127.5))
        print(f"{topo}: {dt} mode: f32 frames {n/t_fr:.4g} frames/s | bytes, fused {n/t_by:.4g}")
    print(f"{topo}: f32 frames {n/t_f32:.4g} frames/s | bytes, two passes {n/t_two:.4g} | bytes, fused {n/t_fused:.4g} "
          f"({256*n/t_fused/1e12:.2f} TB/s of input)", flush=True)
