#!/usr/bin/env python3
"""Whole-forward time of the bf16 / fp8 VT-CNN2 against the per-launch chunk size (frames per mdc_forward call).
Small chunks keep a chunk's bf16 features (21 KB/frame) inside the 256 MiB Infinity Cache between the conv kernel's
stores and dense1's loads; large chunks amortise launches and the conv kernel's per-launch weight load."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames
n = 1 << 20
x = synthetic_frames(n, seed=2016, device="cuda:0")
probs = torch.empty((n, 11), dtype=torch.float32, device=x.device); labels = torch.empty((n,), dtype=torch.int32, device=x.device)
for dtype in sys.argv[1:] or ["bf16"]:
    m = VTCNN2.synthetic(Topology.vtcnn2(11), seed=2016, device=0, dtype=dtype)
    for chunk in (4096, 8192, 12288, 16384, 24576, 32768, 65536, 131072, 8192, 65536):
        for _ in range(2): m.forward_device(x, probs, labels, batch_size=chunk)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): m.forward_device(x, probs, labels, batch_size=chunk)
        torch.cuda.synchronize(); el = (time.perf_counter() - t) / 5
        m.set_profiling(True)
        for _ in range(2): m.forward_device(x, probs, labels, batch_size=chunk)
        torch.cuda.synchronize()
        prof = {k: round(v[0] / 2, 2) for k, v in m.read_profile().items()}
        m.set_profiling(False)
        print(f"{dtype} chunk {chunk}: {el*1e3:.2f} ms per 2^20 frames -> {n/el:.4g} frames/s   kernels (ms, with event overhead) {prof}", flush=True)
