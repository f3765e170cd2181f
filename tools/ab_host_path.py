#!/usr/bin/env python3
"""ON THE GPU BOX: bench.py's host-resident leg (numpy in, numpy out through mdc_predict_host) for two BUILDS of the library
in interleaved child processes of one run -- `tools/ab_host_path.py tools/ab_prev.so [rounds = 3]`."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prev = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from modulationdetectioncnn_amd import _cabi
if sys.argv[1] != "current":
    _cabi.LIB_PATHS["product"] = sys.argv[1]
import bench
r = bench.run_host_path(0, steps=6, warmup=2)
print("ROWS", json.dumps([{k: (round(v, 1) if isinstance(v, float) else v) for k, v in row.items() if k != "frames"} for row in r["rows"]]))
''' % ROOT
for rnd in range(rounds):
    for tag, lib in (("prev", prev), ("current", "current")):
        r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("ROWS")]
        if r.returncode != 0 or not line:
            print(r.stderr[-1500:]); sys.exit(1)
        for row in json.loads(line[0][5:]):
            print(f"round {rnd} {tag:8s} {row['workload']:24s} driver {row['streaming_driver_frames_per_s']:.4g}  pinned {row['caller_pinned_frames_per_s']:.4g}  ms {row['ms']}", flush=True)
