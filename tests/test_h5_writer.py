"""formats/h5mini.write_keras_h5: the file ModelCheckpoint / model.save leaves behind (cnn.py:143-147; CNN.ipynb cell 8).

Three independent readers judge a written file:
  * this package's own reader (bit-identical round trip of every tensor);
  * the REAL libhdf5 of this image (h5dump from /opt/conda, HDF5 1.10.6 -- the library h5py wraps; it is part of the
    image, not of the reference): a file written from the contents of a bundled checkpoint must dump to the SAME text as
    that checkpoint -- same groups, same datatypes down to string padding and character set, same attribute values,
    same data;
  * libhdf5 through ctypes (H5Fopen / H5Dopen2 / H5Dread), as the reference's h5py would call it;
  * libhdf5's own tools: h5diff against the bundled checkpoint (no differences; one changed bit is found), h5repack + h5diff
    round trips, h5stat's accounting.
No reference file travels: the tests that read /root/reference skip where it is absent (the GPU box)."""
import ctypes as C
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from modulationdetectioncnn_amd import VTCNN2, Topology, synthetic_weights
from modulationdetectioncnn_amd.formats.h5mini import H5File, load_keras_h5, write_keras_h5
from modulationdetectioncnn_amd.topology import keras_model_config, keras_training_config
from tests.conftest import H5_NAMES

H5DUMP = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
LIBHDF5 = "/opt/conda/lib/libhdf5.so.103"


def _optimizer_of(f: H5File, weighted):
    rd = lambda p: f.read("optimizer_weights/Adam/" + p)
    return {"iterations": int(rd("iter:0")),
            "m": [(rd(f"{l}/kernel/m:0"), rd(f"{l}/bias/m:0")) for l in weighted],
            "v": [(rd(f"{l}/kernel/v:0"), rd(f"{l}/bias/v:0")) for l in weighted]}


def _rewrite(ref_path, out_path):
    ck = load_keras_h5(ref_path)
    f = H5File(ref_path)
    topo = Topology.from_keras_config(ck.model_config)
    names = [l["config"]["name"] for l in ck.model_config["config"]["layers"][1:]]
    weighted = [n for n in ck.layer_names if ck.weights[n]]
    w = [(ck.weights[n][0][1], ck.weights[n][1][1]) for n in weighted]
    write_keras_h5(out_path, topo, w, optimizer=_optimizer_of(f, weighted), layer_names=names,
                   model_name=ck.model_config["config"]["name"])
    return topo, names, f


@pytest.mark.parametrize("name", H5_NAMES)
def test_json_attributes_are_the_bundled_files_byte_for_byte(reference_dir, name):
    f = H5File(os.path.join(reference_dir, name + ".wts.h5"))
    cfg = json.loads(f.root.attrs["model_config"])
    topo = Topology.from_keras_config(cfg)
    names = [l["config"]["name"] for l in cfg["config"]["layers"][1:]]
    assert json.dumps(keras_model_config(topo, topo.keras_layer_names(names), cfg["config"]["name"])) == f.root.attrs["model_config"]
    assert json.dumps(keras_training_config({})) == f.root.attrs["training_config"]


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="no h5dump in this image")
@pytest.mark.parametrize("name", H5_NAMES)
def test_libhdf5_sees_the_same_file_as_the_reference_checkpoint(reference_dir, tmp_path, name):
    ref = os.path.join(reference_dir, name + ".wts.h5")
    out = str(tmp_path / "rewritten.h5")
    _rewrite(ref, out)
    dumps = []
    for p in (ref, out):
        r = subprocess.run([H5DUMP, p], capture_output=True, text=True)
        assert r.returncode == 0 and not r.stderr.strip(), r.stderr
        dumps.append(r.stdout.splitlines()[1:])          # line 0 names the file
    assert len(dumps[0]) > 100
    assert dumps[0] == dumps[1]


@pytest.mark.parametrize("name", H5_NAMES)
def test_tree_matches_the_reference_checkpoint_by_our_reader(reference_dir, tmp_path, name):
    ref = os.path.join(reference_dir, name + ".wts.h5")
    out = str(tmp_path / "rewritten.h5")
    _rewrite(ref, out)
    a, b = H5File(ref), H5File(out)
    wa, wb = dict(a.walk()), dict(b.walk())
    assert list(wa) == list(wb)                                   # same paths in the same (sorted) order
    assert a.root.attrs == b.root.attrs
    for p in wa:
        assert wa[p].is_dataset == wb[p].is_dataset
        assert wa[p].shape == wb[p].shape
        assert set(wa[p].attrs) == set(wb[p].attrs)
        for k, v in wa[p].attrs.items():
            assert np.array_equal(np.asarray(v), np.asarray(wb[p].attrs[k])), (p, k)
        if wa[p].is_dataset:
            assert a.read(p).dtype == b.read(p).dtype and np.array_equal(a.read(p), b.read(p)), p


@pytest.mark.parametrize("kind", ["deployed3", "deployed10", "cnnpy", "vtcnn2"])
def test_round_trip_bit_identical(tmp_path, kind):
    """write -> load_keras_h5 -> the same bits, for every topology (the 13-layer VT-CNN2 needs two symbol-table nodes per
    group), with and without optimizer state."""
    topo = {"deployed3": Topology.deployed(3), "deployed10": Topology.deployed(10), "cnnpy": Topology.cnnpy(10, 10, 5),
            "vtcnn2": Topology.vtcnn2(11)}[kind]
    w = synthetic_weights(topo, seed=5, bias_scale=0.1)
    rng = np.random.default_rng(1)
    opt = {"iterations": 12345678901, "m": [(rng.standard_normal(k.shape).astype(np.float32), rng.standard_normal(b.shape).astype(np.float32)) for k, b in w],
           "v": [(rng.random(k.shape).astype(np.float32), rng.random(b.shape).astype(np.float32)) for k, b in w]}
    for optimizer in (None, opt):
        path = str(tmp_path / f"{kind}.h5")
        write_keras_h5(path, topo, w, optimizer=optimizer, adam=dict(lr=2e-3))
        ck = load_keras_h5(path)
        assert Topology.from_keras_config(ck.model_config) == topo
        assert ck.keras_version == "2.4.0" and ck.backend == "tensorflow"
        m = VTCNN2.from_h5(path)
        for (k, b), (k2, b2) in zip(w, m.get_weights()):
            assert k2.dtype == np.float32 and np.array_equal(k, k2) and np.array_equal(b, b2)
        f = H5File(path)
        assert ("optimizer_weights" in f.root.children) == (optimizer is not None)
        assert json.loads(f.root.attrs["training_config"])["optimizer_config"]["config"]["learning_rate"] == float(np.float32(2e-3))
        if optimizer is not None:
            weighted = [n for n in ck.layer_names if ck.weights[n]]
            back = _optimizer_of(f, weighted)
            assert back["iterations"] == opt["iterations"]
            for mv in ("m", "v"):
                for (k, b), (k2, b2) in zip(opt[mv], back[mv]):
                    assert np.array_equal(k, k2) and np.array_equal(b, b2)
            assert f.get("optimizer_weights").attrs["weight_names"][0] == "Adam/iter:0"


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="no h5dump in this image")
@pytest.mark.parametrize("kind", ["deployed10", "cnnpy", "vtcnn2"])
def test_libhdf5_walks_every_topologys_file(tmp_path, kind):
    """h5dump (the real libhdf5) walks the whole file of every topology without a complaint and finds every dataset with its
    shape -- the 13-layer VT-CNN2 spreads /model_weights over two symbol-table nodes under one B-tree node."""
    topo = {"deployed10": Topology.deployed(10), "cnnpy": Topology.cnnpy(10, 10, 5), "vtcnn2": Topology.vtcnn2(11)}[kind]
    w = synthetic_weights(topo, seed=2)
    opt = {"iterations": 7, "m": w, "v": w}
    path = str(tmp_path / "f.h5")
    write_keras_h5(path, topo, w, optimizer=opt)
    r = subprocess.run([H5DUMP, "-H", path], capture_output=True, text=True)
    assert r.returncode == 0 and not r.stderr.strip(), r.stderr
    names = [n for _, n in topo.keras_layer_names()]
    for role, lname in topo.keras_layer_names():
        assert f'GROUP "{lname}"' in r.stdout
    for (ks, bs), lname in zip(topo.layer_shapes, [n for role, n in topo.keras_layer_names() if role in ("conv", "dense")]):
        dims = ", ".join(str(d) for d in ks)
        assert f"DATASPACE  SIMPLE {{ ( {dims} ) / ( {dims} ) }}" in r.stdout, (lname, dims)
    assert r.stdout.count('DATASET "kernel:0"') == len(topo.layer_shapes)
    assert r.stdout.count('DATASET "m:0"') == 2 * len(topo.layer_shapes) and 'DATASET "iter:0"' in r.stdout
    assert len(names) == len(set(names))


@pytest.mark.skipif(not os.path.exists(LIBHDF5), reason="no libhdf5 in this image")
def test_libhdf5_reads_the_tensors_through_its_c_api(tmp_path):
    """H5Fopen + H5Dopen2 + H5Dread -- the calls behind h5py's `f['model_weights/...'][()]` in Keras' load_weights."""
    topo = Topology.cnnpy(10, 10, 5)
    w = synthetic_weights(topo, seed=8, bias_scale=0.2)
    path = str(tmp_path / "t4.h5")
    write_keras_h5(path, topo, w)
    L = C.CDLL(LIBHDF5)
    hid = C.c_int64
    L.H5open.restype = C.c_int
    L.H5Fopen.restype, L.H5Fopen.argtypes = hid, [C.c_char_p, C.c_uint, hid]
    L.H5Dopen2.restype, L.H5Dopen2.argtypes = hid, [hid, C.c_char_p, hid]
    L.H5Dread.restype, L.H5Dread.argtypes = C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]
    L.H5Dget_storage_size.restype, L.H5Dget_storage_size.argtypes = C.c_uint64, [hid]
    L.H5Dclose.argtypes = L.H5Fclose.argtypes = [hid]
    assert L.H5open() >= 0
    f32 = hid.in_dll(L, "H5T_NATIVE_FLOAT_g").value
    fid = L.H5Fopen(path.encode(), 0, 0)
    assert fid >= 0
    try:
        for lname, (k, b) in zip(("conv2d", "dense", "dense_1"), w):
            for ds, arr in ((f"/model_weights/{lname}/{lname}/kernel:0", k), (f"/model_weights/{lname}/{lname}/bias:0", b)):
                d = L.H5Dopen2(fid, ds.encode(), 0)
                assert d >= 0, ds
                assert L.H5Dget_storage_size(d) == arr.nbytes
                out = np.empty(arr.shape, np.float32)
                assert L.H5Dread(d, f32, 0, 0, 0, out.ctypes.data) >= 0
                L.H5Dclose(d)
                assert np.array_equal(out, arr), ds
    finally:
        L.H5Fclose(fid)


def test_fresh_session_layer_names():
    assert [n for _, n in Topology.deployed(3).keras_layer_names()] == \
        ["reshape", "zero_padding2d", "conv2d", "flatten", "dense", "activation", "reshape_1"]
    assert [n for _, n in Topology.cnnpy().keras_layer_names()] == \
        ["reshape", "zero_padding2d", "conv2d", "flatten", "dense", "dense_1", "activation", "reshape_1"]
    with pytest.raises(ValueError):
        Topology.deployed(3).keras_layer_names(["a", "b"])


def test_shape_mismatch_is_refused(tmp_path):
    topo = Topology.deployed(3)
    w = synthetic_weights(topo)
    with pytest.raises(ValueError):
        write_keras_h5(str(tmp_path / "x.h5"), topo, [(w[0][0], w[0][1]), (w[1][0][:-1], w[1][1])])
    with pytest.raises(ValueError):
        write_keras_h5(str(tmp_path / "x.h5"), topo, w[:1])


# ---------------------------------------------------------------------------------------------------------------------
# libhdf5's own tools as judges (HDF5 1.10.6 of this image; absent on a box without /opt/conda they skip):
#   h5diff    object-by-object, attribute-by-attribute, element-by-element comparison of two files
#   h5repack  reads EVERYTHING through the library and writes it again: a file it can repack is a file h5py can read
#   h5stat    walks all metadata (object headers, B-trees, heaps) and accounts for the file's bytes
# ---------------------------------------------------------------------------------------------------------------------
def _tool(name):
    p = shutil.which(name) or os.path.join("/opt/conda/bin", name)
    return p if os.path.exists(p) else None


@pytest.mark.skipif(_tool("h5diff") is None, reason="no h5diff in this image")
@pytest.mark.parametrize("name", H5_NAMES)
def test_h5diff_finds_no_difference_to_the_reference_checkpoint(reference_dir, tmp_path, name):
    ref = os.path.join(reference_dir, name + ".wts.h5")
    out = str(tmp_path / "rewritten.h5")
    _rewrite(ref, out)
    r = subprocess.run([_tool("h5diff"), "-c", ref, out], capture_output=True, text=True)      # rc 0: no differences; -c: list what cannot be compared
    assert r.returncode == 0 and "not comparable" not in r.stdout and not r.stderr.strip(), r.stdout + r.stderr
    # ... and it does see one when there is one (a single weight changed in the last bit)
    ck = load_keras_h5(ref)
    topo = Topology.from_keras_config(ck.model_config)
    weighted = [n for n in ck.layer_names if ck.weights[n]]
    w = [(ck.weights[n][0][1].copy(), ck.weights[n][1][1].copy()) for n in weighted]
    w[1][0][5, 1] = np.nextafter(w[1][0][5, 1], np.float32(9))
    names = [l["config"]["name"] for l in ck.model_config["config"]["layers"][1:]]
    write_keras_h5(out, topo, w, optimizer=_optimizer_of(H5File(ref), weighted), layer_names=names, model_name=ck.model_config["config"]["name"])
    r = subprocess.run([_tool("h5diff"), ref, out], capture_output=True, text=True)
    assert r.returncode == 1 and "1 differences found" in r.stdout, r.stdout


@pytest.mark.skipif(_tool("h5repack") is None or _tool("h5diff") is None or _tool("h5stat") is None, reason="no HDF5 tools in this image")
@pytest.mark.parametrize("kind", ["deployed3", "deployed10", "cnnpy", "vtcnn2"])
def test_libhdf5_repacks_and_accounts_for_every_topologys_file(tmp_path, kind):
    topo = {"deployed3": Topology.deployed(3), "deployed10": Topology.deployed(10), "cnnpy": Topology.cnnpy(10, 10, 5),
            "vtcnn2": Topology.vtcnn2(11)}[kind]
    w = synthetic_weights(topo, seed=2, bias_scale=0.1)
    for opt in (None, {"iterations": 41, "m": w, "v": w}):
        src, dst = str(tmp_path / "w.h5"), str(tmp_path / "repacked.h5")
        if os.path.exists(dst):
            os.remove(dst)
        write_keras_h5(src, topo, w, optimizer=opt)
        r = subprocess.run([_tool("h5repack"), src, dst], capture_output=True, text=True)
        assert r.returncode == 0 and not r.stderr.strip(), r.stderr
        r = subprocess.run([_tool("h5diff"), src, dst], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        back = load_keras_h5(dst)                                     # and our reader takes libhdf5's rewrite of our file
        for (k, b), lname in zip(w, [n for n in back.layer_names if back.weights[n]]):
            assert np.array_equal(back.weights[lname][0][1], k) and np.array_equal(back.weights[lname][1][1], b)
        r = subprocess.run([_tool("h5stat"), src], capture_output=True, text=True)
        assert r.returncode == 0 and not r.stderr.strip(), r.stderr
        assert f"# of unique datasets: {2 * len(topo.layer_shapes) + (1 + 4 * len(topo.layer_shapes) if opt else 0)}" in r.stdout, r.stdout[:600]
