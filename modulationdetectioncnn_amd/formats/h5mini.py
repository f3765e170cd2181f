"""Minimal read-only HDF5 reader for Keras 2.4 full-model checkpoints.

Replaces the ``h5py`` half of ``model.load_weights(filepath)`` (reference
cnn.py:147, CNN.ipynb cell 8).  h5py is not installed here nor on the GPU box,
so this reads exactly the subset of HDF5 the five bundled ``*.wts.h5`` files
use: superblock v0, 8-byte offsets/lengths,
old-style groups (v1 B-tree -> SNOD -> local heap), v1 object headers with
continuation blocks, contiguous f32 datasets, v1 attributes holding fixed
strings, arrays of fixed strings, or variable-length strings in a global heap.

Nothing from the file is executed; it is parsed as bytes only.
"""
from __future__ import annotations

import json
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


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
        # superblock v0: root group symbol-table entry starts at byte 56
        root_hdr = struct.unpack_from("<Q", b, 64)[0]
        self.root = self._read_object("/", root_hdr)

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
        return tuple(self._u("Q", doff + 8 * i)[0] for i in range(rank))

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
        count = int(np.prod(shape)) if shape else 1
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
            if b[node:node + 4] == b"SNOD":
                nsym = self._u("H", node + 6)[0]
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
        dt = ds.dtype.numpy()
        count = int(np.prod(ds.shape)) if ds.shape else 1
        if ds.compact is not None:
            arr = np.frombuffer(ds.compact, dtype=dt, count=count)
        else:
            if ds.data_addr in (None, _UNDEF):
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
    mw = f.get("model_weights") if "model_weights" in root.children else root
    layer_names = [str(s) for s in mw.attrs.get("layer_names", [])]
    weights: Dict[str, List[Tuple[str, np.ndarray]]] = {}
    for lname in layer_names:
        grp = mw.children[lname]
        wnames = grp.attrs.get("weight_names", [])
        if isinstance(wnames, str):
            wnames = [wnames]
        base = "model_weights/" if mw is not root else ""
        weights[lname] = [(str(w), f.read(f"{base}{lname}/{w}")) for w in wnames]
    return KerasCheckpoint(
        path=path,
        keras_version=str(root.attrs.get("keras_version", "")),
        backend=str(root.attrs.get("backend", "")),
        model_config=json.loads(cfg_raw),
        layer_names=layer_names,
        weights=weights,
    )
