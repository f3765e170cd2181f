"""Format decoders: Keras .h5 (h5mini) and Q6.12 text tables (q612).  CPU only.

The reference's data files are read from /root/reference when present (build container);
on the GPU box these tests skip and the derived fixtures in tests/golden stand in."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, H5_NAMES
from modulationdetectioncnn_amd.formats import q612
from modulationdetectioncnn_amd.formats.h5mini import H5File, H5FormatError, load_keras_h5
from modulationdetectioncnn_amd.topology import Topology

PAIRS = [  # which txt export belongs to which checkpoint
    ("2convmodrecnets_CNN2_0.5", "12.14.weights.txt", ("conv_kernel", "conv_bias", "dense_bias")),
    ("2convmodrecnets_CNN2_0.5", "12.15.denseWeights.txt", ("dense_kernel",)),
    ("3convmodrecnets_CNN2_0.5", "12.15.latestWeights.txt", ("conv_kernel", "conv_bias", "dense_kernel", "dense_bias")),
    ("4convmodrecnets_CNN2_0.5", "am.fm.qpsk.txt", ("conv_kernel", "conv_bias", "dense_kernel", "dense_bias")),
    ("5convmodrecnets_CNN2_0.5", "am.fm.8psk.txt", ("conv_kernel", "conv_bias", "dense_kernel", "dense_bias")),
    ("convmodrecnets_CNN2_0.5", "DenseWeights1.txt", ("dense_kernel",)),
]


@pytest.mark.parametrize("name", H5_NAMES)
def test_h5_matches_fixture(reference_dir, name):
    ck = load_keras_h5(os.path.join(reference_dir, name + ".wts.h5"))
    assert ck.keras_version == "2.4.0" and ck.backend == "tensorflow"
    z = np.load(os.path.join(GOLDEN, "weights", name + ".npz"))
    tensors = [a for l in ck.layer_names for _, a in ck.weights[l]]
    for got, key in zip(tensors, ("conv_kernel", "conv_bias", "dense_kernel", "dense_bias")):
        assert got.dtype == np.float32
        np.testing.assert_array_equal(got, z[key])
    topo = Topology.from_keras_config(ck.model_config)
    F = 10 if name == "convmodrecnets_CNN2_0.5" else 3       # the digit prefix is a run counter, not a conv count
    assert topo == Topology.deployed(F, 3)
    assert tensors[0].shape == (1, 2, 1, F) and tensors[2].shape == (258 * F, 3)


def test_h5_known_offsets(reference_dir):
    # contiguous f32 datasets at fixed offsets in 3conv
    f = H5File(os.path.join(reference_dir, "3convmodrecnets_CNN2_0.5.wts.h5"))
    assert f.get("model_weights/conv2d_3/conv2d_3/kernel:0").data_addr == 10336
    assert f.get("model_weights/conv2d_3/conv2d_3/bias:0").data_addr == 10360
    assert f.get("model_weights/dense_3/dense_3/bias:0").data_addr == 10372
    assert f.get("model_weights/dense_3/dense_3/kernel:0").data_addr == 15136
    assert "optimizer_weights" in f.root.children          # present, ignored by the loader


def test_h5_rejects_garbage(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all" * 10)
    with pytest.raises(H5FormatError):
        H5File(str(p))


def test_float2fix_roundtrip_and_bug():
    for v in (0.0, 1.0, -1.0, 3.4700375, -0.00724824, 31.999, -32.0, 2 ** -12, -(2 ** -12)):
        bits = q612.float2fix(v)
        assert len(bits) == 18
        assert q612.bits_to_int(bits) == int(v * 4096)
    # recorded by the reference itself: CNN.ipynb cell 21 prints output_dense[0] = [3.510959 3.1282985 4.310499] and cell 25 their
    # float2fix(x, 18, 12) strings
    for v, bits in ((3.510959, "000011100000101100"), (3.1282985, "000011001000001101"), (4.310499, "000100010011110111")):
        assert q612.float2fix(v) == bits
    # the reference generator's negative-zero bug (CNN.ipynb cell 23): 19 characters
    assert q612.float2fix(-1e-5, bug_compatible=True) == q612.NEGZERO_19
    assert q612.float2fix(-1e-5) == "0" * 18
    assert q612.bits_to_int(q612.NEGZERO_19) == 0 and q612.bits_to_int(q612.NEGZERO_18) == 0
    assert q612.bits_to_int(q612.NEGZERO_19, strict=True) == -(1 << 17)     # Verilog keeps the low 18 bits: -32.0
    assert q612.bits_to_int(q612.NEGZERO_18, strict=True) == -(1 << 16)     # hand-trimmed form: -16.0


def test_parse_grammar_inline():
    text = """* Conv Bias + Weights
        18'd00: data <= 18'b000001011110101110;
        18'd01: data <= 18'b110000000000000000;
18'd111111111111010111 // first class bias
18'b000000000101000011
first table
18'd000: data = 18'b1100000000000000000;
18'd001: data = 18'b111111111111111111;
18'd000: data = 18'b000000000000000001;
"""
    p = q612.parse_text(text)
    assert [len(t) for t in p.tables] == [2, 2, 1]
    assert p.tables[0].rows == [0b000001011110101110, 0] and p.tables[0].negzero_rows == [1]
    assert p.bare == [-41, 323]                 # 18'd typo accepted as a bare token
    assert p.tables[1].rows == [0, -1] and p.tables[1].negzero_rows == [0]
    assert p.tables[2].rows == [1]
    strict = q612.parse_text(text, strict=True)
    assert strict.tables[0].rows[1] == -(1 << 16) and strict.tables[1].rows[0] == -(1 << 17)


@pytest.mark.parametrize("h5name,txt,parts", PAIRS)
def test_txt_weights_equal_h5_to_one_lsb(reference_dir, h5name, txt, parts):
    z = np.load(os.path.join(GOLDEN, "weights", h5name + ".npz"))
    w = q612.load_weights_txt(os.path.join(reference_dir, txt))
    for key in parts:
        got = getattr(w, key)
        assert got is not None, key
        # float2fix truncates toward zero: |txt - h5| < 2**-12 everywhere (bug rows decode to 0 = trunc)
        assert np.abs(got.reshape(z[key].shape) - z[key]).max() <= 2.0 ** -12 + 1e-7, key
    if txt == "12.14.weights.txt":
        assert w.placeholder_dense              # six identical dense tables: not real weights


def test_negzero_inventory(reference_dir):
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    nz = {k: v["negzero"] for k, v in man["txt"].items()}
    # SURVEY.md 8(a) A6 occurrences
    assert nz["12.15.latestWeights.txt"]["dense"] == [5 * 387 + 373]
    assert nz["am.fm.8psk.txt"]["dense"] == [3 * 387 + 230, 5 * 387 + 265, 5 * 387 + 377]
    assert nz["am.fm.qpsk.txt"]["conv"] == [5] and nz["am.fm.qpsk.txt"]["dense"] == [95, 387 + 60, 387 + 158]
    assert nz["12.15.denseWeights.txt"]["dense"] == [55, 366]
    assert len(nz["DenseWeights1.txt"]["dense"]) == 4
    for t in man["txt"]:
        again = q612.load_weights_txt(os.path.join(reference_dir, t))
        assert again.negzero == man["txt"][t]["negzero"]


def test_frames_match_fixture_and_bug_counts(reference_dir):
    meta = json.load(open(os.path.join(GOLDEN, "frames.json")))
    raw = np.load(os.path.join(GOLDEN, "frames.npz"))["raw"]
    assert raw.shape == (16, 2, 128)
    i = 0
    counts = {}
    for fn in sorted(set(n.split("#")[0] for n in meta["names"]), key=[n.split("#")[0] for n in meta["names"]].index):
        ff = q612.load_frames(os.path.join(reference_dir, fn))
        for k in range(ff.raw.shape[0]):
            np.testing.assert_array_equal(ff.raw[k], raw[i])
            assert ff.negzero[k] == meta["negzero_rows"][i]
            counts[fn] = counts.get(fn, 0) + len(ff.negzero[k])
            i += 1
    assert i == 16
    # SURVEY.md 8(a) A7 counts of float2fix bug tokens per file
    assert counts["newTestDataClass2.txt"] == 25 and counts["12.14.testdata.class2.txt"] == 21
    assert counts["12.15testDataClass2.txt"] == 4 and counts["12.15.testDataClass3.txt"] == 3
    assert counts["12.15.testDataClass1.txt"] == 2 and counts["newTestData.txt"] == 2 and counts["12.15.newTestFourth.txt"] == 2
    names = meta["names"]
    # duplicates noted by the survey
    np.testing.assert_array_equal(raw[names.index("newTestData.txt")], raw[names.index("12.15.testDataClass1.txt")])
    np.testing.assert_array_equal(raw[names.index("12.16.testDataYunyun.txt#1")], raw[names.index("12.15.sixtyfourSamples.txt")])
    # 12.15.sixSampleData.txt is ONE frame with 6 non-zero samples (not "6 frames")
    six = raw[names.index("12.15.sixSampleData.txt")]
    assert np.count_nonzero(six) == 6 and np.count_nonzero(six[:, :3]) == 6


def test_unrepaired_bug_tokens_wreck_the_frame(reference_dir):
    strict = q612.load_frames(os.path.join(reference_dir, "newTestDataClass2.txt"), strict=True)
    assert strict.frames.min() <= -16.0
    fixed = q612.load_frames(os.path.join(reference_dir, "newTestDataClass2.txt"))
    assert np.abs(fixed.frames).max() < 0.1


def test_yunyun_recorded_predictions(reference_dir):
    ff = q612.load_frames(os.path.join(reference_dir, "12.16.testDataYunyun.txt"))
    assert ff.predictions[0] == [0.0, 3.1391976, 0.3649335]
    assert ff.predictions[1] == [3.4700375, 2.4710786, 1.3579643]


def test_frame_writer_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((2, 128)) * 5e-3).astype(np.float32)
    x[0, 5] = -1e-5                       # would hit the generator bug
    p = tmp_path / "f.txt"
    p.write_text(q612.dump_frame(x))
    back = q612.load_frames(str(p))
    assert back.negzero == [[]]
    np.testing.assert_array_equal(back.raw[0], np.trunc(x.astype(np.float64) * 4096).astype(np.int32))
    p.write_text(q612.dump_frame(x, bug_compatible=True))
    want = [int(i) for i in np.flatnonzero((x.ravel() < 0) & (x.ravel() > -(2.0 ** -12)))]
    assert 5 in want and q612.load_frames(str(p)).negzero == [want]


def test_weight_writer_roundtrip(tmp_path):
    z = np.load(os.path.join(GOLDEN, "weights", "3convmodrecnets_CNN2_0.5.npz"))
    w = q612.DeployedWeights(3, z["conv_kernel"], z["conv_bias"], z["dense_kernel"], z["dense_bias"])
    p = tmp_path / "w.txt"
    p.write_text(q612.dump_weights_f3(w))
    back = q612.load_weights_f3(str(p))
    for key in ("conv_kernel", "conv_bias", "dense_kernel", "dense_bias"):
        want = np.trunc(z[key].astype(np.float64) * 4096) / 4096
        np.testing.assert_array_equal(getattr(back, key).reshape(z[key].shape), want.astype(np.float32))


def test_sv_rom_equals_latest_weights(reference_dir):
    """cnn_test_latest1.sv ROM rows == 12.15.latestWeights.txt bit for bit."""
    sv = q612.parse_file(os.path.join(reference_dir, "cnn_test_latest1.sv"), strict=True)
    txt = q612.parse_file(os.path.join(reference_dir, "12.15.latestWeights.txt"), strict=True)
    sv_dense = [t.rows for t in sv.tables if len(t) == 387]
    txt_dense = [t.rows for t in txt.tables if len(t) == 387]
    assert len(sv_dense) == 6 and sv_dense == txt_dense


def test_results_file_loader_runs_no_code_from_the_file(tmp_path):
    """load_results used plain pickle.load.  The results tuple of cnn.py:262-264 needs no global at all, so the
    loader refuses every one: a pickle that names a callable (the classic os.system payload, or just a numpy scalar from
    somebody else's writer) raises instead of importing anything; a well-formed file still round-trips."""
    import pickle
    from modulationdetectioncnn_amd import VTCNN2
    good = tmp_path / "results_cnn2_d0.5.dat"
    VTCNN2.save_results(str(good), {-20: 0.25, 0: 0.5, 18: 0.875})
    assert VTCNN2.load_results(str(good)) == ("CNN2", 0.5, {-20: 0.25, 0: 0.5, 18: 0.875})
    marker = tmp_path / "pwned"

    class Payload:
        def __reduce__(self):
            import os
            return (os.system, (f"touch {marker}",))
    evil = tmp_path / "evil.dat"
    evil.write_bytes(pickle.dumps(("CNN2", 0.5, {0: Payload()}), protocol=2))
    with pytest.raises(pickle.UnpicklingError, match="refusing"):
        VTCNN2.load_results(str(evil))
    assert not marker.exists()
    npscalar = tmp_path / "np.dat"
    npscalar.write_bytes(pickle.dumps(("CNN2", 0.5, {0: np.float64(0.5)}), protocol=2))      # needs numpy globals: refused too
    with pytest.raises(pickle.UnpicklingError):
        VTCNN2.load_results(str(npscalar))
    shape = tmp_path / "shape.dat"
    shape.write_bytes(pickle.dumps(["not", "a", "results", "tuple"], protocol=2))
    with pytest.raises(ValueError, match="results file"):
        VTCNN2.load_results(str(shape))
