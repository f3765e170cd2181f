#!/usr/bin/env python3
"""Deployed nets at the reference's precision: the all-VALU f32 kernel (deployed.hip, production) against the variant
with the dense layer on the f32 matrix pipe (deployed_f32m.hip; alternates build, MDC_DEP_F32_MFMA=1), f32 frames and raw
uint8 I/Q.  With --ablate: rebuilds the alternates library with -DMDC_ABLATIONS and times the probes of
deployed_f32m_kernel (results wrong by construction), then restores it."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
from modulationdetectioncnn_amd import VTCNN2, synthetic_frames
G = os.path.join(%r, "tests", "golden", "weights")
VARIANT = os.environ.get("MDC_TOOL_VARIANT", "product")
n = 1 << 20
x = synthetic_frames(n, seed=2016, device="cuda:0")
iq = torch.randint(0, 256, (n * 256,), dtype=torch.uint8, device="cuda:0")
probs = torch.empty((n, 3), dtype=torch.float32, device="cuda"); labels = torch.empty((n,), dtype=torch.int32, device="cuda")
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
for wl in ("deployed3-f32-n2^20", "deployed10-f32-n2^20"):
    m = VTCNN2.from_npz(os.path.join(G, ("3" if wl.startswith("deployed3") else "") + "convmodrecnets_CNN2_0.5.npz"), device=0, _lib_variant=VARIANT)
    el = timed(lambda: m.forward_device(x, probs, labels))
    eu = timed(lambda: m.predict_iq_u8(iq, 0.02 / 127.5))
    print(f"{os.environ.get('TAG','')} {wl}: {n/el:.4g} frames/s ({n*1036/el/1e12:.2f} TB/s = {n*1036/el/8e12:.3f} of HBM peak); raw u8: {n/eu:.4g} frames/s", flush=True)
''' % (ROOT, ROOT)

def run(tag, **env):
    e = dict(os.environ, TAG=tag, **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=600)
    print(r.stdout.strip() or r.stderr[-800:], flush=True)

if "--ablate" in sys.argv:
    from modulationdetectioncnn_amd import build as b
    b.build(variant="alternates", extra_flags=["-DMDC_ABLATIONS"])
    try:
        run("[all-VALU, product]")
        run("[all-VALU: plain fmas in independent phases (probe)]", MDC_TOOL_VARIANT="alternates", MDC_DEP_PHASED=1)
        run("[all-VALU phased, cache-resident frames (probe)]", MDC_TOOL_VARIANT="alternates", MDC_DEP_PHASED=2)
        run("[all-VALU packed, cache-resident frames (probe)]", MDC_TOOL_VARIANT="alternates", MDC_DEP_PHASED=3)
        for abl, what in ((0, "MFMA-dense variant"), (1, "no MFMA (conv + one VALU mul-add)"), (2, "no conv VALU (MFMA on raw samples)"), (3, "neither: streaming + reduction only"),
                          (4, "no block reduction"), (8, "no DMA"), (9, "no DMA, no MFMA"), (11, "no DMA, no MFMA, no conv"), (102, "streaming probe, ring of 2 groups"), (103, "ring of 3")):
            run(f"[f32m abl {abl}: {what}]", MDC_TOOL_VARIANT="alternates", MDC_DEP_F32_MFMA=1, MDC_ABLATE_F32M=abl)
    finally:
        b.build(variant="alternates")
else:
    run("[all-VALU (production)]")
    run("[f32 MFMA-dense variant]", MDC_TOOL_VARIANT="alternates", MDC_DEP_F32_MFMA=1)
