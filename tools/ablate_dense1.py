#!/usr/bin/env python3
"""Timing-only probes of the bf16 dense1 GEMM (MDC_ABLATE_D1=1: all work-groups stream the same A rows,
2: no MFMA, 3: no staging traffic -- on the one-barrier kernel; 10: the phased kernel, 11: phased with A rows from L2).  Results are wrong by construction; only the kernel time is read."""
import os, subprocess, sys, json
sys.path.insert(0, ".")
from modulationdetectioncnn_amd import build as _b
_b.build(force=True, extra_flags=["-DMDC_ABLATIONS"])   # NOTE: rebuild without the flag afterwards
for abl in (sys.argv[1:] or ["0", "1", "2", "3", "10", "11"]):
    env = dict(os.environ, MDC_ABLATE_D1=abl)
    r = subprocess.run([sys.executable, "bench.py", "--no-extras", "--no-cpu-baseline", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print("ABLATE", abl, {k: round(v["ms_per_step"], 2) for k, v in j["kernels"].items()}, "value %.3g" % j["value"], flush=True)
    except Exception as e:
        print("ABLATE", abl, "failed", e, r.stderr[-500:])
