#!/usr/bin/env python3
"""Generate tests/golden/ from the reference's DATA files (run in the build container).

Reads (never executes) under /root/reference:
  * the five Keras .h5 checkpoints            -> golden/weights/<name>.npz   (decoded f32 tensors)
  * the fifteen Q6.12 frame .txt files         -> golden/frames.npz + frames.json
  * the seven Q6.12 weight .txt files          -> golden/weights_txt/<name>.npz
  * CNN.ipynb cell 18 stored output (text)     -> golden/keras_kat.json  (float input + Keras answer)
  * CNN.ipynb cell 19 stored output (text)     -> golden/keras_kat_t2_flat.json  (21 entries of the 10-filter net's Flatten
                                                  output as Keras printed them; the input frame was not printed)
  * the "* prediction:" comments of 12.16.testDataYunyun.txt -> frames.json
  * the three stored model.summary() printouts (CNN.ipynb cell 6 = T1, cnn.ipynb = T4, the DeepSig notebook = T3)
                                               -> golden/summaries.json (layer kinds, output shapes, parameter counts)

Everything written is derived data (arrays / numbers); no reference source text is copied.
Oracle-derived expectations (labels frozen per SURVEY.md 8(c)(5)) are written by
tools/freeze_oracle_labels.py AFTER the oracle passes the Keras known answers.
"""
import glob
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from modulationdetectioncnn_amd.formats import q612            # noqa: E402
from modulationdetectioncnn_amd.formats.h5mini import load_keras_h5   # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

FRAME_FILES = [
    "12.14.testdata.class2.txt", "12.14.testdata.class3.txt", "12.15.newTestFirst.txt",
    "12.15.newTestSecond.txt", "12.15.newTestThird.txt", "12.15.newTestFourth.txt",
    "12.15.sixSampleData.txt", "12.15.sixtyfourSamples.txt", "12.15.testDataClass1.txt",
    "12.15testDataClass2.txt", "12.15.testDataClass3.txt", "12.16.testDataYunyun.txt",
    "newTestData.txt", "newTestDataClass2.txt", "newTestDataClass3.txt",
]
WEIGHT_TXT = ["12.14.weights.txt", "12.15.denseWeights.txt", "12.15.latestWeights.txt",
              "am.fm.8psk.txt", "am.fm.qpsk.txt", "DenseWeights1.txt", "newDenseWeights.txt"]


def cell18():
    nb = json.load(open(os.path.join(REF, "CNN.ipynb")))
    cell = nb["cells"][18]
    text = "".join("".join(o.get("text", "")) for o in cell["outputs"])
    # the printed (1,2,128) array, then the 3-vector on the last line
    head, tail = text.rsplit("]]]", 1)
    nums = [float(t) for t in re.findall(r"-?\d+\.\d*(?:e[-+]?\d+)?|-?\d+\.", head)]
    assert len(nums) == 256, len(nums)
    pred = [float(t) for t in re.findall(r"-?\d+\.\d+", tail)]
    assert len(pred) == 3, pred
    return np.asarray(nums, np.float32).reshape(1, 2, 128), pred, cell.get("execution_count")


def cell19():
    """`print(output_conv_relu[0])` + the element prints of CNN.ipynb cell 19: output_conv_relu = model3.predict(...) is the Flatten
    output (2,580 values: a 10-filter net) of a session whose input frame is not in the notebook.  -> {flat index: value}."""
    nb = json.load(open(os.path.join(REF, "CNN.ipynb")))
    cell = nb["cells"][19]
    text = "".join("".join(o.get("text", "")) for o in cell["outputs"])
    lines = text.split("\n")
    arr = [float(t) for t in re.findall(r"\d+\.\d*(?:e[-+]?\d+)?", lines[0])]            # [a b c ... x y z]: numpy's summary of 2,580 values
    assert len(arr) == 6 and "..." in lines[0], lines[0]
    first = [float(v) for v in lines[lines.index("First 10 Values") + 1: lines.index("next 10")]]
    nxt = [float(v) for v in lines[lines.index("next 10") + 1: lines.index("Last Value")]]
    last = float(lines[lines.index("Last Value") + 1])
    assert len(first) == 10 and len(nxt) == 8, (first, nxt)
    # the summary line prints the same leading values to fewer digits: they must agree with the element prints
    assert all(abs(a - b) < 1e-8 for a, b in zip(arr[:3], first[:3])) and abs(arr[5] - last) < 1e-8
    vals = {i: v for i, v in enumerate(first)}
    vals.update({10 + i: v for i, v in enumerate(nxt)})
    vals.update({2577: arr[3], 2578: arr[4], 2579: last})
    return vals, cell.get("execution_count")


def printed_frames():
    """CNN.ipynb cells 14 and 16 print two frames of the data set in full float precision (`print(X_test[2])`, `print(X_test[3][0])` /
    `[1]`): the only unquantised RML2016.10a frames the reference holds besides cell 18's.  -> (2, 2, 128) float32."""
    nb = json.load(open(os.path.join(REF, "CNN.ipynb")))
    num = r"-?\d+\.\d*(?:e[-+]?\d+)?"
    t14 = "".join("".join(o.get("text", "")) for o in nb["cells"][14]["outputs"])
    t16 = "".join("".join(o.get("text", "")) for o in nb["cells"][16]["outputs"])
    a = [float(t) for t in re.findall(num, t14)]
    i16, q16 = t16.split("Q Data (128)")
    b = [float(t) for t in re.findall(num, i16.split("I Data (128)")[1])] + [float(t) for t in re.findall(num, q16)]
    assert len(a) == 256 and len(b) == 256, (len(a), len(b))
    return np.asarray([a, b], np.float32).reshape(2, 2, 128)


SUMMARIES = {      # topology tag -> (notebook, how the build names it)
    "deployed3": "CNN.ipynb",
    "cnnpy": "cnn.ipynb",
    "vtcnn2": "examples-master/modulation_recognition/RML2016.10a_VTCNN2_example.ipynb",
}
_ROW = re.compile(r"^\S+\s+\((\w+)\)?\s+\((None(?:,\s*\d+)+)\)\s+(\d+)")


def stored_summaries():
    """The model.summary() tables Keras printed into the notebooks' stored outputs: per layer the class name (as
    printed: Keras 2 truncates "ZeroPadding2D" to "ZeroPaddin"), the output shape without the batch axis and the
    parameter count; plus the "Total params" line.  Numerals and class names only -- no notebook text is kept."""
    out = {}
    for tag, rel in SUMMARIES.items():
        nb = json.load(open(os.path.join(REF, rel)))
        found = None
        for ci, cell in enumerate(nb["cells"]):
            for o in cell.get("outputs", []):
                text = "".join(o.get("text", ""))
                if "Layer (type)" in text and "Total params" in text:
                    found = (ci, text)
        assert found, rel
        ci, text = found
        layers = []
        for line in text.splitlines():
            m = _ROW.match(line)
            if m:
                layers.append({"class": m.group(1), "output_shape": [int(t) for t in m.group(2).split(",")[1:]],
                               "params": int(m.group(3))})
        total = int(re.search(r"Total params:\s*([\d,]+)", text).group(1).replace(",", ""))
        assert sum(l["params"] for l in layers) == total, (tag, total)
        out[tag] = {"source": f"{rel} cell {ci} stored output (model.summary())", "layers": layers, "total_params": total}
    return out


def main():
    os.makedirs(os.path.join(OUT, "weights"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "weights_txt"), exist_ok=True)
    manifest = {"h5": {}, "txt": {}}
    for p in sorted(glob.glob(os.path.join(REF, "*.h5"))):
        ck = load_keras_h5(p)
        name = os.path.basename(p).replace(".wts.h5", "")
        arrs = {}
        tensors = [a for l in ck.layer_names for _, a in ck.weights[l]]
        assert len(tensors) == 4
        arrs = dict(conv_kernel=tensors[0], conv_bias=tensors[1], dense_kernel=tensors[2], dense_bias=tensors[3])
        np.savez(os.path.join(OUT, "weights", name + ".npz"), **arrs)
        layers = [(l["class_name"], {k: l["config"].get(k) for k in
                                     ("target_shape", "padding", "filters", "kernel_size", "units", "activation", "data_format")
                                     if k in l["config"]}) for l in ck.layer_configs()]
        manifest["h5"][name] = {"keras_version": ck.keras_version, "backend": ck.backend,
                                "layer_names": ck.layer_names, "layers": layers,
                                "shapes": {k: list(v.shape) for k, v in arrs.items()},
                                "bytes": os.path.getsize(p)}
    for t in WEIGHT_TXT:
        w = q612.load_weights_txt(os.path.join(REF, t))
        arrs = {k: getattr(w, k) for k in ("conv_kernel", "conv_bias", "dense_kernel", "dense_bias") if getattr(w, k) is not None}
        np.savez(os.path.join(OUT, "weights_txt", t[:-4] + ".npz"), **arrs)
        manifest["txt"][t] = {"filters": w.filters, "parts": sorted(arrs), "negzero": w.negzero,
                              "placeholder_dense": w.placeholder_dense}

    names, raws, nzs, preds = [], [], [], []
    for fn in FRAME_FILES:
        ff = q612.load_frames(os.path.join(REF, fn))
        strict = q612.load_frames(os.path.join(REF, fn), strict=True)
        for i in range(ff.raw.shape[0]):
            names.append(fn if ff.raw.shape[0] == 1 else f"{fn}#{i}")
            raws.append(ff.raw[i])
            nzs.append(ff.negzero[i])
            preds.append(ff.predictions[i])
            assert (strict.raw[i].ravel()[ff.negzero[i]] < 0).all()
    np.savez(os.path.join(OUT, "frames.npz"), raw=np.stack(raws).astype(np.int32))
    json.dump({"names": names, "negzero_rows": nzs, "keras_prediction": preds,
               "note": "raw = Q6.12 integers after negative-zero repair; value = raw/4096"},
              open(os.path.join(OUT, "frames.json"), "w"), indent=1)

    x, pred, exe = cell18()
    json.dump({"source": "CNN.ipynb cell 18 stored output (execution_count %s)" % exe,
               "weights": "3convmodrecnets_CNN2_0.5", "tap": "dense (model2 = layers[4].output, post-ReLU, pre-softmax)",
               "input": [float(np.float32(v)) for v in x.ravel()], "keras_dense": pred},
              open(os.path.join(OUT, "keras_kat.json"), "w"))
    np.savez(os.path.join(OUT, "notebook_frames.npz"), frames=printed_frames(),
             source=np.array("CNN.ipynb cells 14 (X_test[2]) and 16 (X_test[3]): stored print output, float32 as printed"))
    vals, exe19 = cell19()
    json.dump({"source": "CNN.ipynb cell 19 stored output (execution_count %s): entries of output_conv_relu[0] = model3.predict(...)[0]" % exe19,
               "weights": "convmodrecnets_CNN2_0.5 (the bundled 10-filter checkpoint: the 21 entries are mutually consistent under ITS conv "
                          "kernel and bias -- three input samples explain all of them, tests/test_oracle_golden.py)",
               "tap": "flat (model3 = layers[3].output: Conv2D + ReLU, flattened channels_last, index h*1290 + w*10 + f)",
               "input": "not printed in the notebook; three of its samples (I[0], I[1], Q[127]) follow from three of the entries",
               "flat_index": sorted(vals), "keras_flat": [vals[i] for i in sorted(vals)]},
              open(os.path.join(OUT, "keras_kat_t2_flat.json"), "w"), indent=1)
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)
    json.dump(stored_summaries(), open(os.path.join(OUT, "summaries.json"), "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
