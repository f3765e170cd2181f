"""Pin the CPU oracle to the reference's own known answers; CPU only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, H5_NAMES, load_deployed_npz
from oracle import oracle_np as O


def _kat():
    k = json.load(open(os.path.join(GOLDEN, "keras_kat.json")))
    return np.asarray(k["input"], np.float32).reshape(1, 2, 128), np.asarray(k["keras_dense"])


def _frames():
    raw = np.load(os.path.join(GOLDEN, "frames.npz"))["raw"]
    meta = json.load(open(os.path.join(GOLDEN, "frames.json")))
    return raw.astype(np.float32) / np.float32(4096), meta


def _flat(w):
    return [a for p in w for a in p]


def test_keras_known_answer_float_input():
    """CNN.ipynb cell 18: model2.predict(newTest1) with 3conv weights -> [3.4700375 2.4710786 1.3579643]."""
    x, want = _kat()
    w = _flat(load_deployed_npz("3convmodrecnets_CNN2_0.5"))
    r32 = O.forward_deployed(x, *w, dtype=np.float32)
    r64 = O.forward_deployed(x, *w, dtype=np.float64)
    assert np.abs(r32["dense"][0] - want).max() < 5e-6       # fp32 tolerance
    assert np.abs(r64["dense"][0] - want).max() < 1e-6
    assert r32["labels"][0] == 0 and r64["labels"][0] == 0   # WBFM
    np.testing.assert_allclose(r64["probs"][0], [0.67148, 0.24728, 0.08124], atol=1e-5)
    assert x[0, :, 64:].max() == 0 and x[0, :, 64:].min() == 0   # samples 64..127 were zeroed in the notebook


def test_keras_known_answers_quantised_input():
    """12.16.testDataYunyun.txt:1-2 and :263-264 (inputs quantised to Q6.12)."""
    x, meta = _frames()
    w = _flat(load_deployed_npz("3convmodrecnets_CNN2_0.5"))
    r = O.forward_deployed(x, *w, dtype=np.float64)
    i0, i1 = meta["names"].index("12.16.testDataYunyun.txt#0"), meta["names"].index("12.16.testDataYunyun.txt#1")
    assert meta["keras_prediction"][i0] == [0.0, 3.1391976, 0.3649335]
    assert np.abs(r["dense"][i0] - np.array(meta["keras_prediction"][i0])).max() < 5e-3
    assert r["labels"][i0] == 1                                    # AM-SSB
    assert np.abs(r["dense"][i1] - np.array(meta["keras_prediction"][i1])).max() < 0.1   # input truncation moves it 3 %
    assert r["labels"][i1] == 0


def test_quantised_kat_frame_is_float2fix_of_float_input():
    from modulationdetectioncnn_amd.formats import q612
    x, _ = _kat()
    xf, meta = _frames()
    i1 = meta["names"].index("12.16.testDataYunyun.txt#1")
    q = np.array([q612.bits_to_int(q612.float2fix(float(v))) for v in x.ravel()]).reshape(2, 128)
    np.testing.assert_array_equal(q, np.load(os.path.join(GOLDEN, "frames.npz"))["raw"][i1])


@pytest.mark.parametrize("name", H5_NAMES)
def test_frozen_oracle_outputs(name):
    """Oracle-derived (not Keras-recorded) dense outputs and labels for every bundled frame."""
    x, meta = _frames()
    fz = json.load(open(os.path.join(GOLDEN, "oracle_frozen.json")))
    assert fz["frames"] == meta["names"]
    w = _flat(load_deployed_npz(name))
    r = O.forward_deployed(x, *w, dtype=np.float64)
    np.testing.assert_allclose(r["dense"], np.array(fz["by_weights"][name]["dense"]), rtol=0, atol=1e-9)
    assert r["labels"].tolist() == fz["by_weights"][name]["labels"]
    r32 = O.forward_deployed(x, *w, dtype=np.float32)
    assert r32["labels"].tolist() == fz["by_weights"][name]["labels"]


def test_survey_label_list_3conv():
    _, meta = _frames()
    fz = json.load(open(os.path.join(GOLDEN, "oracle_frozen.json")))
    got = dict(zip(meta["names"], fz["by_weights"]["3convmodrecnets_CNN2_0.5"]["labels"]))
    want = {"12.15.testDataClass1.txt": 0, "newTestData.txt": 0, "12.15testDataClass2.txt": 1, "12.15.testDataClass3.txt": 2,
            "12.14.testdata.class2.txt": 1, "12.14.testdata.class3.txt": 2, "12.15.newTestFirst.txt": 1,
            "12.15.newTestSecond.txt": 1, "12.15.newTestThird.txt": 1, "12.15.newTestFourth.txt": 1,
            "12.15.sixSampleData.txt": 1, "12.15.sixtyfourSamples.txt": 0, "newTestDataClass2.txt": 1, "newTestDataClass3.txt": 1}
    for k, v in want.items():
        assert got[k] == v, k


def test_argmax_first_max_on_ties():
    p = np.array([[0.2, 0.2, 0.6], [1 / 3, 1 / 3, 1 / 3], [0.1, 0.45, 0.45]])
    assert O.argmax_first(p).tolist() == [2, 0, 1]
    # all-zero Dense+ReLU output (e.g. newTestDataClass2 under 5conv) -> uniform softmax -> label 0
    x, meta = _frames()
    w = _flat(load_deployed_npz("5convmodrecnets_CNN2_0.5"))
    r = O.forward_deployed(x, *w, dtype=np.float64)
    i = meta["names"].index("newTestDataClass2.txt")
    assert r["dense"][i].tolist() == [0.0, 0.0, 0.0] and r["labels"][i] == 0


def test_empty_batch():
    w = _flat(load_deployed_npz("3convmodrecnets_CNN2_0.5"))
    r = O.forward_deployed(np.zeros((0, 2, 128), np.float32), *w)
    assert r["probs"].shape == (0, 3) and r["labels"].shape == (0,)
