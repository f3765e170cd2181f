#!/usr/bin/env python3
"""The training half of the reference's cnn.py (lines 42-153; CNN.ipynb cells 2-9) on this library, call for call:

    X_train / X_test / Y_train / Y_test from the dataset dict                 cnn.py:42-82   -> formats/rml2016.py (or synthetic cells)
    model = Sequential([... Conv2D ... Dense ...]); model.compile(            cnn.py:104-113 -> VTCNN2.synthetic(topology) (glorot_uniform /
        loss='categorical_crossentropy', optimizer='adam')                                      he_normal / zero biases); model.compile(...)
    history = model.fit(X_train, Y_train, batch_size=1024, epochs=100,        cnn.py:135-146 -> the same call (callbacks module of this package)
        validation_data=(X_test, Y_test), callbacks=[ModelCheckpoint(filepath,
        monitor='val_loss', save_best_only=True), EarlyStopping(patience=5)])
    model.load_weights(filepath)                                              cnn.py:147     -> model.load_weights(filepath)  (Keras .h5)
    score = model.evaluate(X_test, Y_test, batch_size=batch_size)             cnn.py:153     -> model.evaluate(...)

The reference trains on RML2016.10a, which is not available here: without --dataset the cells below are synthetic (the
signal-shaped frames of tests/signals.py with a class-dependent level: a separable stand-in, not the dataset).

    python examples/train_like_cnn_py.py [filepath.h5] [--net deployed3|deployed10|cnnpy] [--epochs 100]
    python examples/train_like_cnn_py.py filepath.h5 --dataset RML2016.10a_dict.pkl [--mods WBFM,AM-SSB,GFSK] [--snrs 2,...,18]
           [--train-fraction 0.7] [--seed 2015]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from modulationdetectioncnn_amd import VTCNN2, Topology, callbacks      # noqa: E402   (`keras.callbacks` in the reference)
from modulationdetectioncnn_amd.formats import rml2016        # noqa: E402
from modulationdetectioncnn_amd.training import to_onehot     # noqa: E402


def synthetic_cells(n=18900 + 8100, seed=2015):
    """(X, class indices) shaped like CNN.ipynb cell 2's selection: three classes, 27,000 frames (70 % of them = the 18,900
    training frames cell 5 prints)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from signals import modulated_frames
    x, lab, _snr = modulated_frames(n, seed=seed)
    return (x * np.array([0.4, 1.0, 2.2], np.float32)[lab][:, None, None]).astype(np.float32), lab.astype(np.int64)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("filepath", nargs="?", default="convmodrecnets_CNN2_0.5.wts.h5")      # cnn.py:133
    ap.add_argument("--net", default="deployed3", choices=["deployed3", "deployed10", "cnnpy"])
    ap.add_argument("--epochs", type=int, default=100)          # nb_epoch, cnn.py:122
    ap.add_argument("--batch-size", type=int, default=1024)     # cnn.py:123
    ap.add_argument("--lr", type=float, default=1e-3)           # Keras' Adam default
    ap.add_argument("--dataset")
    ap.add_argument("--mods", default="WBFM,AM-SSB,GFSK")
    ap.add_argument("--snrs", default="2,4,6,8,10,12,14,16,18")
    ap.add_argument("--train-fraction", type=float, default=0.7)
    ap.add_argument("--seed", type=int, default=2015)
    args = ap.parse_args(argv)

    if args.dataset:
        mods = args.mods.split(",")
        ds = rml2016.RML2016.load(args.dataset)                                              # cnn.py:42-45
        X, lbl = ds.select(mods, [int(s) for s in args.snrs.split(",")])                     # cnn.py:49-59
        idx = np.array([mods.index(m) for m, _ in lbl], np.int64)
    else:
        X, idx = synthetic_cells(seed=args.seed)
        mods = ["WBFM", "AM-SSB", "GFSK"]
    train_idx, test_idx = rml2016.split_indices(len(X), args.train_fraction, seed=args.seed)     # cnn.py:66-72
    X_train, X_test = X[train_idx], X[test_idx]
    Y_train, Y_test = to_onehot(idx[train_idx], len(mods)), to_onehot(idx[test_idx], len(mods))  # cnn.py:74-82
    print(X_train.shape, list(X_train.shape[1:]))                                                 # cnn.py:89-90

    topo = {"deployed3": Topology.deployed(3, len(mods)), "deployed10": Topology.deployed(10, len(mods)),
            "cnnpy": Topology.cnnpy(10, 10, len(mods))}[args.net]
    model = VTCNN2.synthetic(topo, seed=args.seed)                                                # cnn.py:104-112 (the initialisers)
    model.compile(loss="categorical_crossentropy", optimizer="adam", lr=args.lr)                  # cnn.py:113
    print(topo.summary())                                                                          # cnn.py:115
    history = model.fit(X_train, Y_train, batch_size=args.batch_size, epochs=args.epochs, verbose=2,        # cnn.py:135-146
                        validation_data=(X_test, Y_test),
                        callbacks=[
                            callbacks.ModelCheckpoint(args.filepath, monitor='val_loss', verbose=0, save_best_only=True, mode='auto'),
                            callbacks.EarlyStopping(monitor='val_loss', patience=5, verbose=0, mode='auto')],
                        seed=args.seed)                                  # (Keras' shuffle is unseeded; this one can be reproduced)
    model.load_weights(args.filepath)                                                              # cnn.py:147
    score = model.evaluate(X_test, Y_test, batch_size=args.batch_size)                             # cnn.py:153
    print(score)
    return model, history, score, (X_test, idx[test_idx])


if __name__ == "__main__":
    main()
