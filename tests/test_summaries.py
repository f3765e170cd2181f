"""The only facts the reference holds about the canonical VT-CNN2 (T3) and cnn.py's literal model (T4) are the
model.summary() tables Keras printed into its notebooks (RML2016.10a_VTCNN2_example.ipynb:190-216: (256,2,130) /
(80,1,132) / 10560, 1024 / 122960 / 2703616 / 2827 parameters, 2,830,427 in all; cnn.ipynb:146-166: (1,4,128) ->
(1,3,10), 2570 / 310 / 55, 2,935 in all; CNN.ipynb cell 6 for the deployed net: 9 / 2325, 2,334).
tools/make_golden.py extracted them into tests/golden/summaries.json (numerals and layer class names).  Here the
build's three statements of each topology -- Topology.layer_shapes (what load_weights takes), the oracle's tap shapes
and the GPU kernels' tap shapes -- are held against those tables."""
import json
import os

import numpy as np
import pytest

from modulationdetectioncnn_amd import Topology, synthetic_frames, synthetic_weights
from oracle import oracle_np as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "summaries.json")))
TOPO = {"deployed3": Topology.deployed(3, 3), "cnnpy": Topology.cnnpy(10, 10, 5), "vtcnn2": Topology.vtcnn2(11)}
PARAM_LAYERS = ("Conv2D", "Convolution2D", "Dense")


def _rows(tag, classes):
    return [l for l in GOLD[tag]["layers"] if l["class"] in classes]


def test_the_fixture_holds_the_numbers_the_notebooks_print():
    """(the numerals VERDICT r3 quotes, so a regenerated fixture that lost a row fails here)"""
    assert GOLD["vtcnn2"]["total_params"] == 2830427 and GOLD["cnnpy"]["total_params"] == 2935 and GOLD["deployed3"]["total_params"] == 2334
    assert [l["params"] for l in _rows("vtcnn2", PARAM_LAYERS)] == [1024, 122960, 2703616, 2827]
    assert [l["params"] for l in _rows("cnnpy", PARAM_LAYERS)] == [2570, 310, 55]
    assert [l["params"] for l in _rows("deployed3", PARAM_LAYERS)] == [9, 2325]
    assert [l["output_shape"] for l in _rows("vtcnn2", ("Convolution2D", "Flatten"))] == [[256, 2, 130], [80, 1, 132], [10560]]
    assert [l["output_shape"] for l in GOLD["cnnpy"]["layers"][:3]] == [[1, 2, 128], [1, 4, 128], [1, 3, 10]]


@pytest.mark.parametrize("tag", sorted(TOPO))
def test_layer_shapes_carry_keras_parameter_counts(tag):
    topo = TOPO[tag]
    want = [l["params"] for l in _rows(tag, PARAM_LAYERS)]
    got = [int(np.prod(k)) + int(np.prod(b)) for k, b in topo.layer_shapes]
    assert got == want
    assert sum(got) == GOLD[tag]["total_params"]
    # synthetic weights (what the headline benchmark runs on) have exactly these shapes
    assert [(k.shape, b.shape) for k, b in synthetic_weights(topo, seed=1)] == [(tuple(k), tuple(b)) for k, b in topo.layer_shapes]
    # the output width of the last parameterised layer is the class count
    assert _rows(tag, PARAM_LAYERS)[-1]["output_shape"] == [topo.classes]


@pytest.mark.parametrize("tag", sorted(TOPO))
def test_the_whole_sequential_definition_matches_the_stored_table(tag):
    """Topology.keras_layers() -- every layer of the definition, weighted or not, in order -- against the stored
    model.summary() table row by row: class (Keras 1 printed "Convolution2D", Keras 2 truncates "ZeroPadding2D" to ten
    characters), output shape, parameter count; and the total.  This is the reference-held pin of the T3 / T4 TOPOLOGIES."""
    rows = TOPO[tag].keras_layers()
    want = GOLD[tag]["layers"]
    assert len(rows) == len(want)
    for (name, shape, params), w in zip(rows, want):
        printed = {"Convolution2D": "Conv2D"}.get(w["class"], w["class"])
        assert name.startswith(printed), (name, w["class"])                 # "ZeroPaddin" is the truncated "ZeroPadding2D"
        assert list(shape) == w["output_shape"], (name, shape, w["output_shape"])
        assert params == w["params"], (name, params, w["params"])
    assert sum(p for _, _, p in rows) == GOLD[tag]["total_params"]
    text = TOPO[tag].summary()
    assert f"Total params: {GOLD[tag]['total_params']:,}" in text and text.count("\n") == len(rows) + 3
    # the weighted rows are the ones load_weights fills, in order
    weighted = [(n, p) for n, _, p in rows if p]
    assert [p for _, p in weighted] == [int(np.prod(k)) + int(np.prod(b)) for k, b in TOPO[tag].layer_shapes]


def _keras_shapes(tag):
    """{our tap name: Keras' printed output shape}"""
    L = GOLD[tag]["layers"]
    convs = [l["output_shape"] for l in L if l["class"] in ("Conv2D", "Convolution2D")]
    flat = next(l["output_shape"] for l in L if l["class"] == "Flatten")
    dens = [l["output_shape"] for l in L if l["class"] == "Dense"]
    if tag == "vtcnn2":
        return {"conv1": convs[0], "conv2": convs[1], "flat": flat, "dense1": dens[0], "logits": dens[1]}
    if tag == "cnnpy":
        return {"conv": convs[0], "flat": flat, "dense1": dens[0], "logits": dens[1]}
    return {"conv": convs[0], "flat": flat, "dense": dens[0]}


@pytest.mark.parametrize("tag", sorted(TOPO))
def test_oracle_taps_have_the_shapes_keras_printed(tag):
    topo = TOPO[tag]
    x = synthetic_frames(2, seed=3)
    kw = {"taps": True} if tag == "vtcnn2" else {}
    res = O.forward(topo.kind, x, synthetic_weights(topo, seed=2), dtype=np.float32, **kw)
    for tap, shape in _keras_shapes(tag).items():
        got = list(res[tap].shape[1:])
        if (tag, tap) == ("vtcnn2", "conv2"):
            assert got == [80, 132] and shape == [80, 1, 132]        # the oracle drops conv2's height-1 axis
        elif (tag, tap) == ("cnnpy", "conv"):
            assert got == [3, 10] and shape == [1, 3, 10]            # ... and the literal model's H = 1
        else:
            assert got == shape, (tap, got, shape)
    assert list(res["probs"].shape[1:]) == GOLD[tag]["layers"][-1]["output_shape"]
    # the padded widths Keras prints are the ones the oracle convolves over: W_out = W_pad - k + 1
    pads = [l["output_shape"] for l in GOLD[tag]["layers"] if l["class"].startswith("ZeroPad")]
    if tag == "vtcnn2":
        assert pads == [[1, 2, 132], [256, 2, 134]] and 132 - 3 + 1 == 130 and 134 - 3 + 1 == 132
    elif tag == "cnnpy":
        assert pads == [[1, 4, 128]]
    else:
        assert pads == [[2, 130, 1]]


@pytest.mark.gpu
@pytest.mark.parametrize("tag,dtype", [("vtcnn2", "f32"), ("vtcnn2", "bf16"), ("cnnpy", "f32"), ("deployed3", "f32")])
def test_gpu_taps_have_the_shapes_keras_printed(tag, dtype):
    """MDC_TAP_* buffers as the kernels fill them: row length and layout per Keras' printed shapes, and (f32) the same
    values as the oracle's tap of that name."""
    from modulationdetectioncnn_amd import VTCNN2
    topo = TOPO[tag]
    w = synthetic_weights(topo, seed=2, bias_scale=0.01)
    m = VTCNN2(topo, device=0, dtype=dtype)
    m.set_weights(w)
    x = synthetic_frames(5, seed=3)
    kw = {"taps": True} if tag == "vtcnn2" else {}
    ref = O.forward(topo.kind, x, w, dtype=np.float64, **kw)
    ks = _keras_shapes(tag)
    names = {"vtcnn2": {"conv": "conv2", "flat": "flat", "hidden": "dense1", "dense": "logits"},
             "cnnpy": {"conv": "conv", "flat": "flat", "hidden": "dense1", "dense": "logits"},
             "deployed3": {"conv": "conv", "flat": "flat", "dense": "dense"}}[tag]
    for tap, oname in names.items():
        got = m.predict(x, tap=tap)
        shape = [d for d in ks[oname] if not (tag == "vtcnn2" and oname == "conv2" and d == 1)]
        assert list(got.shape[1:]) == shape, (tap, got.shape, ks[oname])
        assert int(np.prod(got.shape[1:])) == int(np.prod(ks[oname]))
        if dtype == "f32":
            scale = max(1e-30, float(np.abs(ref[oname]).max()))
            assert float(np.abs(got - ref[oname].reshape(got.shape)).max()) / scale < 2e-5, tap
    assert m.predict(x).shape == (5, GOLD[tag]["layers"][-1]["output_shape"][0])
