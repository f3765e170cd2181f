"""Minimal read-only HDF5 reader for Keras 2.4 full-model checkpoints.

Replaces the ``h5py`` half of ``model.load_weights(filepath)`` (reference
cnn.py:147, CNN.ipynb cell 8).  h5py is not installed here nor on the GPU box,
so this reads exactly the subset of HDF5 the five bundled ``*.wts.h5`` files
use: superblock v0, 8-byte offsets/lengths,
old-style groups (v1 B-tree -> SNOD -> local heap), v1 object headers with
continuation blocks, contiguous f32 datasets, v1 attributes holding fixed
strings, arrays of fixed strings, or variable-length strings in a global heap.

Nothing from the file is executed; it is parsed as bytes only.  A file is untrusted input: whatever is wrong with it --
truncation, offsets into nowhere, cyclic or shared group links, sizes that do not fit the file -- comes out as
``H5FormatError`` (a ``ValueError``) in time and memory bounded by the file's own size, never as a struct / index /
unicode / recursion error, a hang, or an allocation the file's length does not justify (``tests/test_h5_fuzz.py``).
"""
from __future__ import annotations

import json
import math
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF
_MAX_UNALLOCATED = 1 << 26      # elements of a dataset with no storage (reads as zeros) this reader will materialise
_MAX_MEMBERS = 1 << 16          # members of one group
# what a parser of hostile bytes can trip over before its own checks speak; all of it means "this is not a file we read"
_PARSE_ERRORS = (struct.error, IndexError, KeyError, UnicodeDecodeError, OverflowError, TypeError, ValueError, RecursionError,
                 MemoryError)


class H5FormatError(ValueError):
    """The file uses an HDF5 feature outside the supported subset."""


@dataclass
class _Datatype:
    cls: int            # 0 fixed-point, 1 float, 3 string, 9 vlen
    size: int
    signed: bool = True
    vlen_string: bool = False
    big_endian: bool = False

    def numpy(self) -> np.dtype:
        end = ">" if self.big_endian else "<"
        if self.cls == 1:
            return np.dtype(f"{end}f{self.size}")
        if self.cls == 0:
            return np.dtype(f"{end}{'i' if self.signed else 'u'}{self.size}")
        if self.cls == 3:
            return np.dtype(f"S{self.size}")
        raise H5FormatError(f"no numpy dtype for HDF5 class {self.cls}")


@dataclass
class H5Object:
    """A group or a dataset; ``children`` is empty for datasets."""
    name: str
    addr: int
    attrs: Dict[str, object] = field(default_factory=dict)
    children: Dict[str, "H5Object"] = field(default_factory=dict)
    shape: Optional[Tuple[int, ...]] = None
    dtype: Optional[_Datatype] = None
    data_addr: Optional[int] = None
    data_size: Optional[int] = None
    compact: Optional[bytes] = None

    @property
    def is_dataset(self) -> bool:
        return self.dtype is not None and self.shape is not None


class H5File:
    def __init__(self, path: str):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        self.path = path
        b = self.buf
        if b[:8] != _SIG:
            raise H5FormatError("not an HDF5 file")
        if b[8] != 0:
            raise H5FormatError(f"superblock version {b[8]} unsupported (need 0)")
        if b[13] != 8 or b[14] != 8:
            raise H5FormatError("only 8-byte offsets/lengths supported")
        self._seen: set = set()      # object headers and B-tree nodes already walked: a second visit is a cycle or a hard link
        try:
            # superblock v0: root group symbol-table entry starts at byte 56
            root_hdr = struct.unpack_from("<Q", b, 64)[0]
            self.root = self._read_object("/", root_hdr)
        except H5FormatError:
            raise
        except _PARSE_ERRORS as e:
            raise H5FormatError(f"{path}: corrupt or truncated HDF5 structure ({type(e).__name__}: {e})") from e

    # -- low level ---------------------------------------------------------
    def _u(self, fmt: str, off: int):
        return struct.unpack_from("<" + fmt, self.buf, off)

    def _messages(self, addr: int):
        """Yield (type, flags, payload_offset, size) over a v1 object header."""
        ver, _, nmsg, _refc, hsize = self._u("BBHII", addr)
        if ver != 1:
            raise H5FormatError(f"object header version {ver} unsupported")
        blocks = [(addr + 16, hsize)]
        seen = 0
        while blocks and seen < nmsg:
            off, length = blocks.pop(0)
            end = off + length
            while off + 8 <= end and seen < nmsg:
                mtype, msize, mflags = self._u("HHB", off)
                payload = off + 8
                seen += 1
                if mtype == 0x10:
                    c_off, c_len = self._u("QQ", payload)
                    blocks.append((c_off, c_len))
                else:
                    yield mtype, mflags, payload, msize
                off = payload + msize

    def _parse_datatype(self, off: int) -> Tuple[_Datatype, int]:
        cv, b0, _b1, _b2, size = self._u("BBBBI", off)
        cls, ver = cv & 0x0F, cv >> 4
        if ver not in (1, 2, 3):
            raise H5FormatError(f"datatype version {ver}")
        if cls == 1:      # float: 12 bytes of properties
            return _Datatype(1, size, big_endian=bool(b0 & 1)), 8 + 12
        if cls == 0:      # fixed point: 4 bytes of properties
            return _Datatype(0, size, signed=bool(b0 & 0x08), big_endian=bool(b0 & 1)), 8 + 4
        if cls == 3:      # fixed-length string
            return _Datatype(3, size), 8
        if cls == 9:      # variable length; base type follows
            is_string = (b0 & 0x0F) == 1
            base, blen = self._parse_datatype(off + 8)
            return _Datatype(9, size, vlen_string=is_string or base.cls == 3), 8 + blen
        raise H5FormatError(f"datatype class {cls} unsupported")

    def _parse_dataspace(self, off: int) -> Tuple[int, ...]:
        ver, rank, flags = self._u("BBB", off)
        if ver == 1:
            doff = off + 8
        elif ver == 2:
            doff = off + 4
        else:
            raise H5FormatError(f"dataspace version {ver}")
        return tuple(int(self._u("Q", doff + 8 * i)[0]) for i in range(rank))

    def _global_heap_object(self, coll_addr: int, index: int) -> bytes:
        b = self.buf
        if b[coll_addr:coll_addr + 4] != b"GCOL":
            raise H5FormatError("bad global heap collection")
        csize = self._u("Q", coll_addr + 8)[0]
        off, end = coll_addr + 16, coll_addr + csize
        while off + 16 <= end:
            idx, _ref, _res, osize = self._u("HHIQ", off)
            if idx == 0:
                break
            if idx == index:
                return b[off + 16: off + 16 + osize]
            off += 16 + ((osize + 7) & ~7)
        raise H5FormatError(f"global heap object {index} not found")

    def _parse_attribute(self, off: int) -> Tuple[str, object]:
        ver, _, nsize, tsize, ssize = self._u("BBHHH", off)
        if ver != 1:
            raise H5FormatError(f"attribute version {ver}")
        pad = lambda n: (n + 7) & ~7
        p = off + 8
        name = self.buf[p:p + nsize].split(b"\0", 1)[0].decode("utf-8")
        p += pad(nsize)
        dt, _ = self._parse_datatype(p)
        p += pad(tsize)
        shape = self._parse_dataspace(p) if ssize else ()
        p += pad(ssize)
        count = math.prod(shape) if shape else 1
        if count * max(dt.size, 1) > len(self.buf):
            raise H5FormatError(f"attribute {name!r}: {count} elements do not fit the file")
        if dt.cls == 9:
            if not dt.vlen_string:
                raise H5FormatError("only vlen strings supported")
            vals = []
            for i in range(count):
                _ln, gaddr, gidx = struct.unpack_from("<IQI", self.buf, p + 16 * i)
                vals.append(self._global_heap_object(gaddr, gidx).decode("utf-8"))
            value: object = vals if shape else vals[0]
        elif dt.cls == 3:
            raw = np.frombuffer(self.buf, dtype=dt.numpy(), count=count, offset=p)
            strs = [s.split(b"\0", 1)[0].decode("utf-8") for s in raw.tolist()]
            value = strs if shape else strs[0]
        else:
            arr = np.frombuffer(self.buf, dtype=dt.numpy(), count=count, offset=p)
            value = arr.reshape(shape).copy() if shape else arr[0].item()
        return name, value

    def _group_entries(self, btree: int, heap: int) -> List[Tuple[str, int]]:
        b = self.buf
        if b[heap:heap + 4] != b"HEAP":
            raise H5FormatError("bad local heap")
        heap_data = self._u("Q", heap + 24)[0]
        out: List[Tuple[str, int]] = []

        def name_at(o: int) -> str:
            s = heap_data + o
            return b[s:b.index(b"\0", s)].decode("utf-8")

        def walk(node: int) -> None:
            if ("n", node) in self._seen:
                raise H5FormatError("a group B-tree node is linked twice (cycle)")
            self._seen.add(("n", node))
            if b[node:node + 4] == b"SNOD":
                nsym = self._u("H", node + 6)[0]
                if len(out) + nsym > _MAX_MEMBERS:
                    raise H5FormatError("too many members in one group")
                for i in range(nsym):
                    e = node + 8 + 40 * i
                    noff, hdr = self._u("QQ", e)
                    out.append((name_at(noff), hdr))
                return
            if b[node:node + 4] != b"TREE":
                raise H5FormatError("bad group B-tree node")
            ntype, _level, used = self._u("BBH", node + 4)
            if ntype != 0:
                raise H5FormatError("expected a group B-tree")
            for i in range(used):
                child = self._u("Q", node + 24 + 8 + 16 * i)[0]
                walk(child)

        walk(btree)
        return out

    def _read_object(self, name: str, addr: int) -> H5Object:
        if ("o", addr) in self._seen:
            raise H5FormatError(f"object {name!r} is linked twice (cycle or hard link: not in a Keras save)")
        self._seen.add(("o", addr))
        obj = H5Object(name=name, addr=addr)
        symtab = None
        for mtype, _flags, off, size in self._messages(addr):
            if mtype == 0x11:
                symtab = self._u("QQ", off)
            elif mtype == 0x01:
                obj.shape = self._parse_dataspace(off)
            elif mtype == 0x03:
                obj.dtype, _ = self._parse_datatype(off)
            elif mtype == 0x08:
                ver, cls = self._u("BB", off)
                if ver != 3:
                    raise H5FormatError(f"data layout version {ver}")
                if cls == 1:
                    obj.data_addr, obj.data_size = self._u("QQ", off + 2)
                elif cls == 0:
                    n = self._u("H", off + 2)[0]
                    obj.compact = self.buf[off + 4: off + 4 + n]
                else:
                    raise H5FormatError("chunked datasets unsupported")
            elif mtype == 0x0B:
                raise H5FormatError("filtered datasets unsupported")
            elif mtype == 0x0C:
                k, v = self._parse_attribute(off)
                obj.attrs[k] = v
        if symtab is not None:
            for cname, chdr in self._group_entries(*symtab):
                obj.children[cname] = self._read_object(cname, chdr)
        return obj

    # -- public ------------------------------------------------------------
    def get(self, path: str) -> H5Object:
        node = self.root
        for part in [p for p in path.split("/") if p]:
            if part not in node.children:
                raise KeyError(f"{path!r}: no member {part!r} (have {sorted(node.children)})")
            node = node.children[part]
        return node

    def read(self, path: str) -> np.ndarray:
        ds = self.get(path)
        if not ds.is_dataset:
            raise KeyError(f"{path!r} is not a dataset")
        try:
            dt = ds.dtype.numpy()
        except TypeError as e:
            raise H5FormatError(f"{path!r}: element size {ds.dtype.size} is no numpy type") from e
        count = math.prod(ds.shape) if ds.shape else 1
        if ds.compact is not None:
            if count * dt.itemsize > len(ds.compact):
                raise H5FormatError(f"{path!r}: compact data shorter than its shape")
            arr = np.frombuffer(ds.compact, dtype=dt, count=count)
        else:
            if ds.data_addr in (None, _UNDEF):
                if count > _MAX_UNALLOCATED:
                    raise H5FormatError(f"{path!r}: {count} unallocated elements")
                return np.zeros(ds.shape, dtype=dt.newbyteorder("="))
            if ds.data_addr + count * dt.itemsize > len(self.buf):
                raise H5FormatError(f"{path!r}: data runs past end of file")
            arr = np.frombuffer(self.buf, dtype=dt, count=count, offset=ds.data_addr)
        return arr.reshape(ds.shape).astype(dt.newbyteorder("="), copy=True)

    def walk(self, node: Optional[H5Object] = None, prefix: str = ""):
        node = node or self.root
        for name, child in node.children.items():
            p = f"{prefix}/{name}"
            yield p, child
            yield from self.walk(child, p)


@dataclass
class KerasCheckpoint:
    """What ``load_weights`` needs from a Keras 2.4 full-model ``.h5``."""
    path: str
    keras_version: str
    backend: str
    model_config: dict
    layer_names: List[str]
    weights: Dict[str, List[Tuple[str, np.ndarray]]]   # layer -> [(weight_name, array)]

    def layer_configs(self) -> List[dict]:
        return list(self.model_config.get("config", {}).get("layers", []))


def load_keras_h5(path: str) -> KerasCheckpoint:
    """Parse a Keras checkpoint: topology JSON + ordered per-layer weights.

    Follows what Keras ``load_weights`` does (cnn.py:147): layers in the order
    of ``/model_weights`` attr ``layer_names``, each layer's tensors in the
    order of its ``weight_names`` attr.  ``/optimizer_weights`` is ignored.
    """
    f = H5File(path)
    root = f.root
    cfg_raw = root.attrs.get("model_config")
    if cfg_raw is None:
        raise H5FormatError("no model_config attribute: not a Keras full-model save")
    try:
        mw = f.get("model_weights") if "model_weights" in root.children else root
        layer_names = [str(s) for s in np.atleast_1d(mw.attrs.get("layer_names", [])).tolist()]
        weights: Dict[str, List[Tuple[str, np.ndarray]]] = {}
        for lname in layer_names:
            grp = mw.children[lname]
            wnames = grp.attrs.get("weight_names", [])
            if isinstance(wnames, str):
                wnames = [wnames]
            base = "model_weights/" if mw is not root else ""
            weights[lname] = [(str(w), f.read(f"{base}{lname}/{w}")) for w in wnames]
        model_config = json.loads(cfg_raw)
        if not isinstance(model_config, dict):
            raise H5FormatError("model_config is not a JSON object")
    except H5FormatError:
        raise
    except _PARSE_ERRORS as e:
        raise H5FormatError(f"{path}: not a consistent Keras save ({type(e).__name__}: {e})") from e
    return KerasCheckpoint(
        path=path,
        keras_version=str(root.attrs.get("keras_version", "")),
        backend=str(root.attrs.get("backend", "")),
        model_config=model_config,
        layer_names=layer_names,
        weights=weights,
    )


# ======================================================================================================================
# Writer: the file ModelCheckpoint(filepath, ...) leaves behind (cnn.py:143-147; CNN.ipynb cell 8) -- a Keras 2.4
# full-model save -- so that weights trained HERE flow back into the reference's pipeline (its model.load_weights, then
# float2fix and the ROM tables).  Same subset of HDF5 as the reader, the way libhdf5 1.10 lays the bundled files out:
# superblock v0, old-style groups (local heap + v1 B-tree + symbol-table nodes, names sorted), v1 object headers,
# contiguous little-endian datasets, v1 attributes (fixed strings, or variable-length strings in one global heap).
# tests/test_h5_writer.py holds a written file against the real libhdf5 of this image (h5dump: same tree, attributes and
# data as the reference's own file) and against the reader above.
# ======================================================================================================================
_LEAF_K, _INTERNAL_K = 4, 16          # symbol-table node = 2*4 entries; B-tree node = 2*16 children (libhdf5's defaults)


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


# datatype messages (version 1), as libhdf5 encodes the types h5py hands it
_DT_F32 = bytes.fromhex("11201f00040000000000200017080017" "7f000000")
_DT_F64 = bytes.fromhex("11203f000800000000004000340b0034" "ff030000")
_DT_I64 = bytes.fromhex("100800000800000000004000")


def _dt_fixed_string(size: int) -> bytes:
    return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, size)          # class 3, null-padded, ASCII


def _dt_vlen_string(utf8: bool) -> bytes:
    return struct.pack("<BBBBI", 0x19, 0x01, 0x01 if utf8 else 0x00, 0, 16) + bytes.fromhex("100000000100000000000800")


def _dataspace(shape: Optional[Tuple[int, ...]]) -> bytes:
    if shape is None or len(shape) == 0:
        return struct.pack("<BBBB4x", 1, 0, 0, 0)                 # scalar
    dims = b"".join(struct.pack("<Q", d) for d in shape)
    return struct.pack("<BBBB4x", 1, len(shape), 1, 0) + dims + dims


class _Node:
    def __init__(self, name: str):
        self.name = name
        self.attrs: List[Tuple[str, bytes, bytes, bytes]] = []      # (name, datatype, dataspace, data)
        self.children: Dict[str, "_Node"] = {}
        self.array: Optional[np.ndarray] = None                     # datasets only

    def group(self, name: str) -> "_Node":
        return self.children.setdefault(name, _Node(name))

    def dataset(self, name: str, array: np.ndarray) -> None:
        node = _Node(name)
        node.array = array
        self.children[name] = node


class _Writer:
    def __init__(self):
        self.buf = bytearray(96)            # superblock, written last
        self.vlen: List[bytes] = []
        self.gcol_addr = 96                 # the global heap collection follows the superblock

    def alloc(self, data: bytes) -> int:
        addr = len(self.buf)
        self.buf += _pad8(data)
        return addr

    # -- global heap: every variable-length string of the file in ONE collection ---------------------------------
    def vlen_ref(self, s: str) -> bytes:
        self.vlen.append(s.encode("utf-8"))
        return struct.pack("<IQI", len(self.vlen[-1]), self.gcol_addr, len(self.vlen))

    def gcol_blob(self) -> bytes:
        body = b"".join(struct.pack("<HHIQ", i + 1, 0, 0, len(s)) + _pad8(s) for i, s in enumerate(self.vlen))
        size = max(4096, (16 + len(body) + 16 + 4095) // 4096 * 4096)      # libhdf5's minimum collection is 4 KiB
        free = size - 16 - len(body)                                         # object 0 = the free space, its own header included
        blob = b"GCOL" + struct.pack("<B3xQ", 1, size) + body + struct.pack("<HHIQ", 0, 0, 0, free)
        return blob + b"\0" * (size - len(blob))

    # -- messages ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _msg(mtype: int, flags: int, data: bytes) -> bytes:
        data = _pad8(data)
        return struct.pack("<HHB3x", mtype, len(data), flags) + data

    def _attr_msg(self, name: str, dt: bytes, ds: bytes, data: bytes) -> bytes:
        nm = name.encode("ascii") + b"\0"
        return self._msg(0x0C, 4, struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + data)

    def _header(self, msgs: List[bytes]) -> int:
        body = b"".join(msgs)
        return self.alloc(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)

    # -- objects ----------------------------------------------------------------------------------------------------
    def write(self, node: _Node) -> Tuple[int, int, int]:
        """Post-order: children first.  Returns (object header address, B-tree address, heap address); the last two are 0
        for a dataset."""
        attrs = [self._attr_msg(*a) for a in node.attrs]
        if node.array is not None:
            a = node.array
            dt = {np.dtype("<f4"): _DT_F32, np.dtype("<i8"): _DT_I64}[a.dtype]
            raw = a.tobytes()
            data_addr = self.alloc(raw)
            msgs = [self._msg(0x01, 0, _dataspace(a.shape)), self._msg(0x03, 1, dt), self._msg(0x05, 1, bytes([2, 2, 2, 1, 0, 0, 0, 0])),
                    self._msg(0x08, 0, struct.pack("<BBQQ", 3, 1, data_addr, len(raw)))]
            return self._header(msgs + attrs), 0, 0
        kids = sorted(node.children.values(), key=lambda c: c.name.encode("ascii"))      # libhdf5 orders by strcmp
        placed = [(c, *self.write(c)) for c in kids]
        # local heap: "" at offset 0, then the names; no free block (free-list head = 1, libhdf5's H5HL_FREE_NULL)
        seg, offs = bytearray(8), []
        for c in kids:
            offs.append(len(seg))
            seg += _pad8(c.name.encode("ascii") + b"\0")
        seg_addr = self.alloc(bytes(seg))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(seg), 1, seg_addr))
        # symbol-table nodes of <= 2*_LEAF_K entries, one leaf B-tree node over them
        per = 2 * _LEAF_K
        snods, keys = [], [0]
        for s in range(0, max(len(kids), 1), per):
            part = placed[s:s + per]
            ents = b""
            for j, (c, hdr, bt, hp) in enumerate(part):
                cache = 1 if c.array is None else 0
                ents += struct.pack("<QQII", offs[s + j], hdr, cache, 0) + (struct.pack("<QQ", bt, hp) if cache else bytes(16))
            ents += bytes(40 * (per - len(part)))
            snods.append(self.alloc(b"SNOD" + struct.pack("<BBH", 1, 0, len(part)) + ents))
            keys.append(offs[s + len(part) - 1] if part else 0)
        if len(snods) > 2 * _INTERNAL_K:
            raise H5FormatError("too many members for a one-level group B-tree")
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), _UNDEF, _UNDEF) + struct.pack("<Q", keys[0])
        for child, key in zip(snods, keys[1:]):
            tree += struct.pack("<QQ", child, key)
        tree += bytes(24 + 8 * (2 * _INTERNAL_K + 1) + 8 * 2 * _INTERNAL_K - len(tree))
        bt_addr = self.alloc(tree)
        hdr = self._header([self._msg(0x11, 0, struct.pack("<QQ", bt_addr, heap_addr))] + attrs)
        return hdr, bt_addr, heap_addr

    def finish(self, root: _Node) -> bytes:
        """Every vlen_ref() has been taken by now (attributes are built before this call), so the collection's size is known
        and it goes right behind the superblock, where vlen_ref() said it would be."""
        assert len(self.buf) == 96 == self.gcol_addr
        self.buf += self.gcol_blob()
        hdr, bt, hp = self.write(root)
        sb = _SIG + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", _LEAF_K, _INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, _UNDEF, len(self.buf), _UNDEF)
        sb += struct.pack("<QQII", 0, hdr, 1, 0) + struct.pack("<QQ", bt, hp)
        assert len(sb) == 96
        self.buf[0:96] = sb
        return bytes(self.buf)


def _attr_strings(w: _Writer, node: _Node, name: str, values: List[str]) -> None:
    """h5py's encoding of a list of str attribute: fixed-length, null-padded ASCII; an EMPTY list becomes float64 (0,)."""
    if not values:
        node.attrs.append((name, _DT_F64, _dataspace((0,)), b""))
        return
    enc = [v.encode("ascii") for v in values]
    size = max(len(e) for e in enc)
    node.attrs.append((name, _dt_fixed_string(size), _dataspace((len(enc),)), b"".join(e.ljust(size, b"\0") for e in enc)))


def _attr_vlen(w: _Writer, node: _Node, name: str, value: str, utf8: bool) -> None:
    node.attrs.append((name, _dt_vlen_string(utf8), _dataspace(None), w.vlen_ref(value)))


KERAS_VERSION, KERAS_BACKEND = "2.4.0", "tensorflow"      # what the five bundled files say of themselves


def write_keras_h5(path: str, topology, weights, optimizer: Optional[dict] = None, adam: Optional[dict] = None,
                   layer_names: Optional[Dict[str, str]] = None, model_name: str = "sequential") -> None:
    """Write a Keras 2.4 full-model checkpoint: what ``ModelCheckpoint(filepath, ...)`` (cnn.py:143) saves and
    ``model.load_weights(filepath)`` (cnn.py:147) reads back.

    topology: modulationdetectioncnn_amd.Topology; weights: [(kernel, bias)] per weighted layer in Keras' layouts;
    optimizer: {'iterations': int, 'm': [(k, b)], 'v': [(k, b)]} -> the /optimizer_weights group (omitted when None, as
    Keras omits it for an optimizer that has not stepped); adam: {'lr', 'beta1', 'beta2', 'eps'} for training_config.
    layer_names: Keras' auto-generated names by role (Topology.keras_layer_names()); a fresh session's by default."""
    from ..topology import keras_model_config, keras_training_config      # (formats <- topology only here: writer-side JSON)
    names = topology.keras_layer_names(layer_names)
    shapes = topology.layer_shapes
    if len(weights) != len(shapes):
        raise ValueError(f"expected {len(shapes)} (kernel, bias) pairs, got {len(weights)}")
    f32 = lambda a, shape: _checked(a, shape)
    w = _Writer()
    root = _Node("/")
    cfg_json = json.dumps(keras_model_config(topology, names, model_name))
    train_json = json.dumps(keras_training_config(adam or {}))
    _attr_vlen(w, root, "keras_version", KERAS_VERSION, True)
    _attr_vlen(w, root, "backend", KERAS_BACKEND, True)
    _attr_vlen(w, root, "model_config", cfg_json, False)
    _attr_vlen(w, root, "training_config", train_json, False)
    mw = root.group("model_weights")
    _attr_strings(w, mw, "layer_names", [n for _, n in names])
    _attr_vlen(w, mw, "backend", KERAS_BACKEND, False)
    _attr_vlen(w, mw, "keras_version", KERAS_VERSION, False)
    weighted = [n for role, n in names if role in ("conv", "dense")]
    if len(weighted) != len(shapes):
        raise H5FormatError("layer naming and weight list disagree")
    it = iter(zip(weights, shapes))
    for role, lname in names:
        g = mw.group(lname)
        if role not in ("conv", "dense"):
            _attr_strings(w, g, "weight_names", [])
            continue
        (k, b), (ks, bs) = next(it)
        _attr_strings(w, g, "weight_names", [f"{lname}/kernel:0", f"{lname}/bias:0"])
        inner = g.group(lname)
        inner.dataset("kernel:0", f32(k, ks))
        inner.dataset("bias:0", f32(b, bs))
    if optimizer is not None:
        ow = root.group("optimizer_weights")
        wn = ["Adam/iter:0"] + [f"Adam/{l}/{t}/{mv}:0" for mv in ("m", "v") for l in weighted for t in ("kernel", "bias")]
        _attr_strings(w, ow, "weight_names", wn)
        ad = ow.group("Adam")
        ad.dataset("iter:0", np.asarray(int(optimizer["iterations"]), dtype="<i8"))
        for mv in ("m", "v"):
            if len(optimizer[mv]) != len(shapes):
                raise ValueError(f"optimizer[{mv!r}] needs one (kernel, bias) pair per weighted layer")
            for lname, (k, b), (ks, bs) in zip(weighted, optimizer[mv], shapes):
                lg = ad.group(lname)
                lg.group("kernel").dataset(f"{mv}:0", f32(k, ks))
                lg.group("bias").dataset(f"{mv}:0", f32(b, bs))
    blob = w.finish(root)
    with open(path, "wb") as fh:
        fh.write(blob)


def _checked(a, shape) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(a), dtype="<f4")
    if tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected a tensor of shape {tuple(shape)}, got {a.shape}")
    return a
