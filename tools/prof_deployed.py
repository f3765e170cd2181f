#!/usr/bin/env python3
"""rocprofv3 driver: a few forwards of the deployed F=3 / F=10 nets on 2^20 frames.
    prof_deployed.py [f32] [bf16] [f16] [fp8] [u8] [q612]     (default: f32 bf16 f16; u8 = the same dtypes on raw uint8 I/Q; q612 = the integer forward)
MDC_DEP_F32_MFMA=1 in the environment selects the f32 variant with the dense layer on the f32 matrix pipe (it lives in the
alternates test build, libmdc_alt.so, which is then the library that is loaded)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulationdetectioncnn_amd import VTCNN2, synthetic_frames
args = [a for a in sys.argv[1:]] or ["f32", "bf16", "f16"]
g = os.path.join(ROOT, "tests", "golden", "weights")
n = 1 << 20
x = synthetic_frames(n, seed=2016, device="cuda:0")
iq = torch.randint(0, 256, (n * 256,), dtype=torch.uint8, device="cuda:0") if "u8" in args else None
for f in ("3convmodrecnets_CNN2_0.5.npz", "convmodrecnets_CNN2_0.5.npz"):
    for dt in ("f32", "bf16", "f16", "fp8"):
        if dt not in args:
            continue
        m = VTCNN2.from_npz(os.path.join(g, f), device=0, dtype=dt, _lib_variant="alternates" if os.environ.get("MDC_DEP_F32_MFMA") == "1" else "product")
        for _ in range(3):
            m.forward_device(x)
            if iq is not None:
                m.predict_iq_u8(iq, 0.02 / 127.5)
            if dt == "f32" and "q612" in args:
                m.predict_q612(x, as_float=False)
torch.cuda.synchronize()
