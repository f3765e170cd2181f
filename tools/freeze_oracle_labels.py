#!/usr/bin/env python3
"""Freeze oracle-derived expectations for the 16 bundled frames.

Run AFTER tests/test_oracle_golden.py passes (the oracle is then pinned to the two Keras
known answers).  For every bundled checkpoint x every bundled frame (negative-zero repaired)
writes the f64 oracle's Dense+ReLU output and first-max label to
tests/golden/oracle_frozen.json.  These are NOT Keras-recorded values: they freeze the pinned
oracle so later changes to it (or to the decoders) are caught.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_np as O   # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
names = ["2convmodrecnets_CNN2_0.5", "3convmodrecnets_CNN2_0.5", "4convmodrecnets_CNN2_0.5",
         "5convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5"]
raw = np.load(os.path.join(G, "frames.npz"))["raw"]
x = raw.astype(np.float64) / 4096.0
meta = json.load(open(os.path.join(G, "frames.json")))
out = {"frames": meta["names"], "by_weights": {}}
for nm in names:
    z = np.load(os.path.join(G, "weights", nm + ".npz"))
    r = O.forward_deployed(x, z["conv_kernel"], z["conv_bias"], z["dense_kernel"], z["dense_bias"], dtype=np.float64)
    out["by_weights"][nm] = {"dense": [[float(v) for v in row] for row in r["dense"]],
                             "labels": [int(v) for v in r["labels"]]}
json.dump(out, open(os.path.join(G, "oracle_frozen.json"), "w"), indent=1)
print({k: v["labels"] for k, v in out["by_weights"].items()})
