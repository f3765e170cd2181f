#!/usr/bin/env python3
"""ON THE GPU BOX -- experiment, no product kernel: would e4m3 FEATURES (and e4m3 dense1 weights, the operands of
v_mfma_scale_f32_16x16x128_f8f6f4) keep the fp8 mode's label floors?  VERDICT r2 item 7: round 2 rejected them on 2,048
NOISE frames in numpy; repeat on 2^16 noise frames AND 2^16 signal-shaped frames (tests/signals.py), with a per-tensor
power-of-two scale.

Emulation: the features are the fp8 mode's own (tap "flat": conv2 on the block-scaled e4m3 MFMA, bf16-rounded), cast to
e4m3 (torch.float8_e4m3fn, RNE, saturating at 448) after a power-of-two scale 2^k; dense1's weights cast the same way
(2^kw, per tensor); the GEMM itself in f32 (the MFMA accumulates in f32), bias + ReLU, dense2 and softmax in f32 as the
head does.  Labels are held against the exact-f32 KERNELS' labels, frame by frame -- the quantity
tests/test_label_agreement_gpu.py bounds (floors: 0.985 on noise frames, 0.990 on signal frames for the fp8 mode).
k is swept from the best fit (feature absmax -> 448 / 2) downwards: a static scale derived from the stated input range
cannot know the absmax and must leave headroom."""
import json, math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_frames, synthetic_weights
from signals import modulated_frames

torch.backends.cuda.matmul.allow_tf32 = False
N, CH = 1 << 16, 4096
out = {}
for classes in (11, 3):
    topo = Topology.vtcnn2(classes)
    w = synthetic_weights(topo, seed=2016)
    mf, m8 = VTCNN2(topo, dtype="f32"), VTCNN2(topo, dtype="fp8")
    mf.set_weights(w); m8.set_weights(w)
    W1 = torch.from_numpy(w[2][0]).cuda(); b1 = torch.from_numpy(w[2][1]).cuda()
    W2 = torch.from_numpy(w[3][0]).cuda(); b2 = torch.from_numpy(w[3][1]).cuda()
    kw = math.floor(math.log2(224.0 / float(W1.abs().max())))
    W1q = (W1 * 2.0 ** kw).to(torch.float8_e4m3fn).float() * 2.0 ** -kw
    W1b = W1.to(torch.bfloat16).float()
    for tag, x in (("noise", synthetic_frames(N, seed=2016, device="cuda")), ("signal", torch.from_numpy(modulated_frames(N, seed=2016)[0]).cuda())):
        lab32 = mf.predict_classes(x)
        lab8 = m8.predict_classes(x)
        fmax = 0.0
        for s in range(0, N, CH):
            fmax = max(fmax, float(m8.predict(x[s:s + CH], tap="flat").abs().max()))
        kbest = math.floor(math.log2(224.0 / fmax))
        res = {"fp8 mode today (bf16 features)": float((lab8 == lab32).float().mean()), "feature absmax": fmax, "k_best": kbest}
        for dk in (0, -2, -4, -6):
            k = kbest + dk
            agree_q = agree_qw = agree_b = 0
            for s in range(0, N, CH):
                f = m8.predict(x[s:s + CH], tap="flat")
                fq = (f * 2.0 ** k).to(torch.float8_e4m3fn).float() * 2.0 ** -k
                def labels(feat, Wm):
                    hid = torch.relu(feat @ Wm + b1)
                    p = torch.softmax(hid @ W2 + b2, dim=1)
                    return p.argmax(dim=1).int()
                agree_q += int((labels(fq, W1b) == lab32[s:s + CH]).sum())          # e4m3 features, bf16 weights (not an MFMA type pair; for reference)
                agree_qw += int((labels(fq, W1q) == lab32[s:s + CH]).sum())         # e4m3 features AND weights: what the scaled MFMA would compute
                if dk == 0:
                    agree_b += int((labels(f, W1b) == lab32[s:s + CH]).sum())       # the emulation's own baseline: bf16 features and weights
            res[f"k = k_best{dk:+d}: e4m3 features, bf16 weights"] = agree_q / N
            res[f"k = k_best{dk:+d}: e4m3 features, e4m3 weights"] = agree_qw / N
            if dk == 0:
                res["emulation baseline: bf16 features, bf16 weights"] = agree_b / N
        out[f"C={classes} {tag}"] = res
        print(f"C={classes} {tag}", json.dumps(res), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r3_exp_fp8_features.json"), "w"), indent=1)
