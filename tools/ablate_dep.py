#!/usr/bin/env python3
"""Timing-only ablations of the fast deployed kernel (MDC_ABLATE_DEP bit 1: no w=0 capture, bit 2: no reduce-scatter)."""
import os, subprocess, sys, json
sys.path.insert(0, ".")
from modulationdetectioncnn_amd import build as _b
_b.build(force=True, extra_flags=["-DMDC_ABLATIONS"])
for abl in (sys.argv[1:] or ["0", "1", "2", "3"]):
    env = dict(os.environ, MDC_ABLATE_DEP=abl)
    r = subprocess.run([sys.executable, "bench.py", "--workload", "deployed3-f32-n2^20", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print("ABLATE_DEP", abl, "value %.3g frac %.3f" % (j["value"], j["roofline"]["frac"]), flush=True)
    except Exception as e:
        print("ABLATE_DEP", abl, "failed", e, r.stderr[-500:])
