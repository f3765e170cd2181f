"""formats/h5mini's reader on hostile bytes.  A checkpoint is a file somebody hands the library (model.load_weights(filepath),
cnn.py:147): a damaged one must come out as H5FormatError (a ValueError) -- in bounded time, without an allocation the file's own
length does not justify -- or load; never a struct / index / unicode / recursion error, never a hang.  Seeded mutations of a
file this package wrote (no reference file needed), plus the hand-made cases a random flip rarely finds: links that form a
cycle, a group B-tree that points at itself, a dataset with no storage and an absurd shape, sizes past the end of the file."""
import signal
import struct

import numpy as np
import pytest

from modulationdetectioncnn_amd import Topology, synthetic_weights
from modulationdetectioncnn_amd.formats.h5mini import H5File, H5FormatError, load_keras_h5, write_keras_h5


@pytest.fixture(scope="module")
def good(tmp_path_factory):
    topo = Topology.deployed(3)
    w = synthetic_weights(topo, seed=1)
    p = str(tmp_path_factory.mktemp("fz") / "good.h5")
    write_keras_h5(p, topo, w, optimizer={"iterations": 3, "m": w, "v": w})
    return open(p, "rb").read(), topo, w


class _Deadline:
    def __init__(self, seconds):
        self.seconds = seconds

    def __enter__(self):
        def boom(*_a):
            raise TimeoutError("the reader did not come back")
        self.old = signal.signal(signal.SIGALRM, boom)
        signal.alarm(self.seconds)

    def __exit__(self, *exc):
        signal.alarm(0)
        signal.signal(signal.SIGALRM, self.old)
        return False


def _load(tmp_path, data: bytes):
    """load -> Topology -> touch every tensor; H5FormatError / ValueError are the ONLY exceptions allowed out."""
    p = str(tmp_path / "m.h5")
    with open(p, "wb") as fh:
        fh.write(data)
    with _Deadline(20):
        try:
            ck = load_keras_h5(p)
        except H5FormatError:
            return None
        try:
            Topology.from_keras_config(ck.model_config)
        except ValueError:
            return None
        return sum(float(np.nan_to_num(a).sum()) for n in ck.layer_names for _, a in ck.weights[n])


def test_seeded_mutations_only_ever_raise_the_format_error(good, tmp_path):
    base, _topo, _w = good
    structural = 6000            # superblock, heaps, B-trees, object headers and the attributes' strings all lie below this offset
    assert len(base) > structural
    rng = np.random.default_rng(2016)
    loaded = refused = 0
    for it in range(1500):
        b = bytearray(base)
        kind = it % 4
        if kind == 0:                                                 # a few flipped bytes in the structures
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, structural))] = int(rng.integers(0, 256))
        elif kind == 1:                                               # truncation anywhere
            b = b[:int(rng.integers(0, len(b)))]
        elif kind == 2:                                               # an 8-byte field (an offset or a length) replaced
            p = int(rng.integers(0, structural))
            b[p:p + 8] = rng.integers(0, 256, 8, dtype=np.uint8).tobytes()
        else:                                                         # ... by a plausible small or huge value
            p = int(rng.integers(0, structural)) & ~7
            values = [0, 1, 8, len(base) - 1, len(base), 1 << 31, 1 << 40, (1 << 64) - 1]
            b[p:p + 8] = struct.pack("<Q", values[int(rng.integers(0, len(values)))])
        r = _load(tmp_path, bytes(b))
        loaded += r is not None
        refused += r is None
    assert refused > 300 and loaded > 100, (loaded, refused)          # the mutations do reach the parser, and harmless ones still load


def _find_all(b: bytes, sig: bytes):
    out, i = [], b.find(sig)
    while i >= 0:
        out.append(i)
        i = b.find(sig, i + 1)
    return out


def test_a_group_btree_that_points_at_itself_is_refused(good, tmp_path):
    base, *_ = good
    b = bytearray(base)
    tree = _find_all(base, b"TREE")[0]
    b[tree + 24 + 8: tree + 24 + 16] = struct.pack("<Q", tree)        # first child = the node itself
    assert _load(tmp_path, bytes(b)) is None
    with pytest.raises(H5FormatError, match="cycle|twice"):
        p = str(tmp_path / "c.h5")
        open(p, "wb").write(bytes(b))
        H5File(p)


def test_a_member_that_links_back_to_the_root_is_refused(good, tmp_path):
    base, *_ = good
    b = bytearray(base)
    root_hdr = struct.unpack_from("<Q", base, 64)[0]
    p = str(tmp_path / "ok.h5")
    open(p, "wb").write(base)
    f = H5File(p)
    btree, _heap = next(f._u("QQ", off) for mtype, _fl, off, _sz in f._messages(root_hdr) if mtype == 0x11)
    snod = f._u("Q", btree + 24 + 8)[0]                                # the root group's one symbol-table node
    assert base[snod:snod + 4] == b"SNOD"
    b[snod + 8 + 8: snod + 8 + 16] = struct.pack("<Q", root_hdr)       # entry 0's object header = the root's own
    p = str(tmp_path / "c.h5")
    open(p, "wb").write(bytes(b))
    with pytest.raises(H5FormatError, match="twice"):
        H5File(p)


def test_unallocated_dataset_with_an_absurd_shape_allocates_nothing(good, tmp_path):
    base, topo, w = good
    p = str(tmp_path / "g.h5")
    open(p, "wb").write(base)
    f = H5File(p)
    ds = f.get("model_weights/dense/dense/kernel:0")
    assert ds.shape == w[1][0].shape
    ds.data_addr, ds.shape = 0xFFFFFFFFFFFFFFFF, (1 << 40, 1 << 20)    # what a header with an undefined address and huge dims parses to
    with pytest.raises(H5FormatError, match="unallocated"):
        f.read("model_weights/dense/dense/kernel:0")
    ds.shape = (4, 3)                                                  # a small one reads as the fill value, as libhdf5 does
    assert np.array_equal(f.read("model_weights/dense/dense/kernel:0"), np.zeros((4, 3), np.float32))
    ds.data_addr, ds.shape = len(base) - 8, (1 << 62, 4)               # count * itemsize overflows 64 bits: python ints do not wrap
    with pytest.raises(H5FormatError, match="past end"):
        f.read("model_weights/dense/dense/kernel:0")


def test_every_prefix_of_the_structures_is_refused_cleanly(good, tmp_path):
    base, *_ = good
    for cut in list(range(0, 2048, 37)) + [len(base) - 1]:
        assert _load(tmp_path, base[:cut]) is None, cut


def test_model_config_that_is_json_but_not_a_model(good):
    for cfg in ({}, {"config": 3}, {"config": {"layers": [1, 2]}}, {"config": {"layers": [{"class_name": "Conv2D"}]}}, [], "x", None):
        with pytest.raises(ValueError):
            Topology.from_keras_config(cfg)
