#!/usr/bin/env python3
"""The evaluation half of the reference's cnn.py (lines 198-264) on this library, call for call:

    score = model.evaluate(X_test, Y_test, batch_size=batch_size)        cnn.py:153   -> VTCNN2.evaluate (mean categorical cross-entropy)
    test_Y_hat = model.predict(X_test, batch_size=batch_size)            cnn.py:198   -> VTCNN2.predict (numpy in, numpy out)
    conf[j, k] += 1 over the test set, row-normalised                     cnn.py:199-216 -> VTCNN2.confusion
    per-SNR confusion matrices and acc[snr] = cor / (cor + ncor)          cnn.py:228-259 -> VTCNN2.accuracy_by_snr (one forward, one launch)
    cPickle.dump(("CNN2", 0.5, acc), open('results_cnn2_d0.5.dat','wb'))  cnn.py:262-264 -> VTCNN2.save_results

The reference evaluates on RML2016.10a, which is not available here: without --dataset the data below is
synthetic and has the reference's shape -- a dict {(modulation, snr): (n, 2, 128) float32}, flattened the way cnn.py:42-75
flattens it.  With the real file, the data half of cnn.py runs first (formats/rml2016.py: a no-code reader of the Python-2
pickle, the cell selection of cnn.py:49-59, the seeded split of cnn.py:66-72) and the evaluation runs on its X_test:

    python examples/evaluate_like_cnn_py.py [weights.h5 | weights.npz] [results.dat]
    python examples/evaluate_like_cnn_py.py weights.h5 results.dat --dataset RML2016.10a_dict.pkl \
           [--mods WBFM,AM-SSB,GFSK] [--snrs 2,4,6,8,10,12,14,16,18] [--train-fraction 0.7] [--seed 2015]
    (defaults: the 3-class selection, 70/30 split and seed 2015 of CNN.ipynb cells 2 and 4, which trained the bundled .h5 files;
     cnn.py itself uses BPSK,GFSK,QAM16,QPSK,WBFM, 0.5 and seed 2016)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from modulationdetectioncnn_amd import VTCNN2, synthetic_frames      # noqa: E402
from modulationdetectioncnn_amd.formats import rml2016                # noqa: E402


def synthetic_dataset(mods=("8PSK", "BPSK", "QAM16"), snrs=range(-4, 20, 2), per_cell=300, seed=7):
    """{(mod, snr): frames}: noise of fixed power plus a per-class tone whose amplitude follows the SNR (enough structure
    for a classifier to have something to separate; the numbers mean nothing beyond exercising the flow)."""
    rng = np.random.default_rng(seed)
    t = np.arange(128, dtype=np.float32)
    data = {}
    for c, mod in enumerate(mods):
        for snr in snrs:
            amp = np.float32(5e-3 * 10 ** (snr / 20))
            x = synthetic_frames(per_cell, seed=int(rng.integers(1 << 30)))
            x[:, 0, :] += amp * np.cos(2 * np.pi * (c + 1) * t / 16, dtype=np.float32)
            x[:, 1, :] += amp * np.sin(2 * np.pi * (c + 1) * t / 16, dtype=np.float32)
            data[(mod, snr)] = x
    return data


def flatten(data):
    """cnn.py:42-75: X stacked cell by cell, lbl = [(mod, snr)] per frame; classes sorted."""
    mods = sorted({m for m, _ in data})
    X, lbl = [], []
    for (mod, snr), frames in sorted(data.items()):
        X.append(frames)
        lbl += [(mod, snr)] * len(frames)
    return np.vstack(X), lbl, mods


def evaluate(model, X_test, lbl, classes, batch_size=1024, results_path=None):
    Y_idx = np.array([classes.index(m) for m, _ in lbl], np.int32)         # Y_test one-hot -> index (cnn.py:205)
    test_SNRs = np.array([s for _, s in lbl])                               # cnn.py:231
    score = model.evaluate(X_test, Y_idx, batch_size=batch_size)            # cnn.py:153 (the mean categorical cross-entropy)
    print(score)                                                            # cnn.py:154
    test_Y_hat = model.predict(X_test, batch_size=batch_size)               # cnn.py:198
    confnorm = model.confusion(X_test, Y_idx, batch_size=batch_size)        # cnn.py:199-216
    acc, conf_by_snr = model.accuracy_by_snr(X_test, Y_idx, test_SNRs)      # cnn.py:228-259
    if results_path:
        VTCNN2.save_results(results_path, acc, tag="CNN2", dr=0.5)         # cnn.py:262-264
    return test_Y_hat, confnorm, acc, conf_by_snr


def test_split_of(path, mods_chosen, snrs_chosen, train_fraction, seed):
    """cnn.py:42-82 up to X_test: (X_test, lbl of the test frames, classes = mods_chosen)."""
    ds = rml2016.RML2016.load(path)                                          # cnn.py:42-45
    X, lbl = ds.select(mods_chosen, snrs_chosen)                             # cnn.py:49-59
    _train_idx, test_idx = rml2016.split_indices(len(X), train_fraction, seed)   # cnn.py:66-72
    return X[test_idx], [lbl[int(i)] for i in test_idx], list(mods_chosen)   # cnn.py:74, 80-82, 91


def main(argv):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("weights", nargs="?", default=os.path.join(ROOT, "tests", "golden", "weights", "3convmodrecnets_CNN2_0.5.npz"))
    ap.add_argument("results", nargs="?", default=None)
    ap.add_argument("--dataset", help="RML2016.10a_dict.dat / .pkl (not bundled with the reference)")
    ap.add_argument("--mods", default="WBFM,AM-SSB,GFSK")
    ap.add_argument("--snrs", default="2,4,6,8,10,12,14,16,18")
    ap.add_argument("--train-fraction", type=float, default=0.7)
    ap.add_argument("--seed", type=int, default=2015)
    args = ap.parse_args(argv[1:])
    model = VTCNN2.from_h5(args.weights) if args.weights.endswith(".h5") else VTCNN2.from_npz(args.weights)
    if args.dataset:
        X_test, lbl, classes = test_split_of(args.dataset, args.mods.split(","), [int(s) for s in args.snrs.split(",")],
                                             args.train_fraction, args.seed)
        if len(classes) != model.topology.classes:
            raise SystemExit(f"{len(classes)} modulations chosen but the model has {model.topology.classes} classes")
    else:
        X_test, lbl, classes = flatten(synthetic_dataset())
    _, confnorm, acc, _ = evaluate(model, X_test, lbl, classes, results_path=args.results)
    np.set_printoptions(precision=3, suppress=True)
    print(confnorm)
    print(classes)
    print({k: round(v, 3) for k, v in acc.items()})


if __name__ == "__main__":
    main(sys.argv)
