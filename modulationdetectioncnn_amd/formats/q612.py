"""Q6.12 fixed-point text tables: the reference's FPGA export format.

The reference only WRITES these files (generator ``float2fix``, CNN.ipynb cell
23-25) and pastes them into ROM ``case`` rows of cnn_test_latest1.sv
(:86-116 frames, :550-563/:685-707/:719-3132 weights); it has no reader.  This
module is the reader (and, for round trips, a bug-free writer).

Row grammar::

    18'd<idx>: data <= 18'b<bits>;      weights      ("<=" or "=")
    18'd<idx>: data = 18'b<bits>;       frames / 10-filter dense dump
    18'b<bits>                          bare token (dense bias; one file
                                        misspells the prefix as 18'd)

Anything else (lines starting with ``*``, ``-->``, ``first table`` ...,
trailing ``// ...``) is commentary.  A value is an 18-bit two's-complement
integer / 4096.  An index that does not increase starts a new table.

``float2fix`` bug: for -2**-12 < v < 0 it emits the 19-character token
``1100000000000000000``; some were hand-trimmed to the 18-character
``110000000000000000`` (= -16.0).  Both mean "negative zero" and decode to 0.0
unless ``strict=True``.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

WIDTH = 18
FRAC = 12
SCALE = float(1 << FRAC)

NEGZERO_19 = "1100000000000000000"
NEGZERO_18 = "110000000000000000"

_ROW = re.compile(r"^\s*18'd(\d+)\s*:\s*data\s*<?=\s*18'b([01]+)\s*;")
_BARE = re.compile(r"^\s*18'[bd]([01]{17,19})(?![01:])")


class Q612FormatError(ValueError):
    pass


def bits_to_int(bits: str, *, strict: bool = False) -> int:
    """Decode one token to the signed 18-bit integer the ROM would hold."""
    if not strict and bits in (NEGZERO_19, NEGZERO_18):
        return 0
    if len(bits) > WIDTH:
        # Verilog keeps the low 18 bits of an over-long sized literal
        bits = bits[-WIDTH:]
    v = int(bits, 2)
    if len(bits) == WIDTH and bits[0] == "1":
        v -= 1 << WIDTH
    return v


def int_to_bits(v: int) -> str:
    if not -(1 << (WIDTH - 1)) <= v < (1 << (WIDTH - 1)):
        raise OverflowError(f"{v} does not fit Q6.12")
    return format(v & ((1 << WIDTH) - 1), f"0{WIDTH}b")


def float_to_int(x: float) -> int:
    """``float2fix`` semantics (CNN.ipynb cell 23): truncate |x|*2**12 toward 0."""
    return int(x * SCALE)  # python int() truncates toward zero, as abs(int(.)) with sign restored


def float2fix(val: float, width: int = WIDTH, precision: int = FRAC, *, bug_compatible: bool = False) -> str:
    """Writer.  ``bug_compatible=True`` reproduces the 19-char negative-zero token."""
    mag = abs(int(val * 2 ** precision))
    if val >= 0:
        return format(mag, f"0{width}b")
    if mag == 0 and not bug_compatible:
        return "0" * width
    if mag == 0:
        return NEGZERO_19
    return format((1 << width) - mag, f"0{width}b")


@dataclass
class Table:
    rows: List[int] = field(default_factory=list)      # decoded signed ints
    indices: List[int] = field(default_factory=list)
    first_line: int = 0
    negzero_rows: List[int] = field(default_factory=list)   # positions within the table

    def as_float(self) -> np.ndarray:
        return np.asarray(self.rows, dtype=np.float64).astype(np.float32) / np.float32(SCALE)

    def __len__(self) -> int:
        return len(self.rows)


@dataclass
class ParsedText:
    tables: List[Table]
    bare: List[int]                 # bare tokens in file order (decoded)
    bare_lines: List[int]
    bare_negzero: List[int]
    comments: List[Tuple[int, str]]


def parse_text(text: str, *, strict: bool = False) -> ParsedText:
    tables: List[Table] = []
    bare: List[int] = []
    bare_lines: List[int] = []
    bare_negzero: List[int] = []
    comments: List[Tuple[int, str]] = []
    cur: Optional[Table] = None
    for ln, line in enumerate(text.splitlines(), start=1):
        m = _ROW.match(line)
        if m:
            idx, bits = int(m.group(1)), m.group(2)
            if len(bits) not in (WIDTH, WIDTH + 1):
                raise Q612FormatError(f"line {ln}: {len(bits)}-bit token")
            if cur is None or (cur.indices and idx <= cur.indices[-1]):
                cur = Table(first_line=ln)
                tables.append(cur)
            if bits in (NEGZERO_19, NEGZERO_18):
                cur.negzero_rows.append(len(cur.rows))
            elif len(bits) != WIDTH:
                raise Q612FormatError(f"line {ln}: 19-bit token that is not the float2fix bug")
            cur.rows.append(bits_to_int(bits, strict=strict))
            cur.indices.append(idx)
            continue
        m = _BARE.match(line)
        if m:
            bits = m.group(1)
            if bits in (NEGZERO_19, NEGZERO_18):
                bare_negzero.append(len(bare))
            bare.append(bits_to_int(bits, strict=strict))
            bare_lines.append(ln)
            cur = None       # a bare token always separates tables
            continue
        if line.strip():
            comments.append((ln, line.strip()))
    return ParsedText(tables, bare, bare_lines, bare_negzero, comments)


def parse_file(path: str, *, strict: bool = False) -> ParsedText:
    with open(path, "r", encoding="utf-8", errors="replace") as fh:
        return parse_text(fh.read(), strict=strict)


# ---------------------------------------------------------------------------
# frames  (A7)
# ---------------------------------------------------------------------------
@dataclass
class FrameFile:
    frames: np.ndarray                     # (n, 2, 128) float32
    raw: np.ndarray                        # (n, 2, 128) int32 Q6.12 integers
    negzero: List[List[int]]               # per frame: row numbers 0..255 that were bug tokens
    predictions: List[Optional[List[float]]]   # "* prediction: [...]" comment before each frame
    notes: List[str]


_PRED = re.compile(r"prediction\s*:\s*\[([^\]]*)\]")


def load_frames(path: str, *, strict: bool = False) -> FrameFile:
    """256 rows per frame: idx 0..127 = I[0..127], 128..255 = Q[0..127]."""
    p = parse_file(path, strict=strict)
    if p.bare:
        raise Q612FormatError(f"{path}: bare tokens in a frame file")
    raws, nz, preds = [], [], []
    for t in p.tables:
        if len(t) != 256 or t.indices != list(range(256)):
            raise Q612FormatError(f"{path}: frame table at line {t.first_line} has {len(t)} rows")
        raws.append(np.asarray(t.rows, dtype=np.int32).reshape(2, 128))
        nz.append(list(t.negzero_rows))
        pred = None
        for ln, c in p.comments:
            if ln < t.first_line:
                m = _PRED.search(c)
                if m:
                    pred = [float(v) for v in m.group(1).split()]
        preds.append(pred)
    # keep only the nearest preceding prediction per frame
    seen = set()
    for i, pr in enumerate(preds):
        key = tuple(pr) if pr else None
        if key in seen and i > 0 and preds[i - 1] is not None and tuple(preds[i - 1]) == key:
            preds[i] = None
        seen.add(key)
    raw = np.stack(raws) if raws else np.zeros((0, 2, 128), np.int32)
    return FrameFile(frames=(raw.astype(np.float32) / np.float32(SCALE)), raw=raw, negzero=nz,
                     predictions=preds, notes=[c for _, c in p.comments])


def dump_frame(frame: np.ndarray, *, bug_compatible: bool = False) -> str:
    """Writer mirroring CNN.ipynb cell 24 (I rows then Q rows, 3-digit index)."""
    flat = np.asarray(frame, dtype=np.float32).reshape(256)
    return "\n".join(f"18'd{i:03d}: data = 18'b{float2fix(float(v), bug_compatible=bug_compatible)};"
                     for i, v in enumerate(flat)) + "\n"


# ---------------------------------------------------------------------------
# weights  (A6)
# ---------------------------------------------------------------------------
@dataclass
class DeployedWeights:
    """Weights of a 1-conv "deployed" net in Keras layout (any part may be None)."""
    filters: int
    conv_kernel: Optional[np.ndarray] = None   # (1, 2, 1, F)  HWIO
    conv_bias: Optional[np.ndarray] = None     # (F,)
    dense_kernel: Optional[np.ndarray] = None  # (258*F, 3)    rows h*129F + w*F + f
    dense_bias: Optional[np.ndarray] = None    # (3,)
    negzero: Dict[str, List[int]] = field(default_factory=dict)
    placeholder_dense: bool = False


def _q(ints: Sequence[int]) -> np.ndarray:
    return (np.asarray(ints, dtype=np.float64) / SCALE).astype(np.float32)


def load_weights_f3(path: str, *, strict: bool = False) -> DeployedWeights:
    """F=3 export: [conv table 9] [3 bare dense-bias tokens] [6 tables x 387].

    Conv table: per filter f rows ``[K[0,0,0,f], K[0,1,0,f], b[f]]``.  Dense
    tables in order (class0-I, class0-Q, class1-I, ...); in-table idx =
    ``f*129 + w``  ->  Keras row ``h*387 + w*3 + f``, column = class
    (cnn_test_latest1.sv:264-336 reads them this way).  Files holding only a
    subset (``12.15.denseWeights.txt``) give None for the missing parts.
    """
    p = parse_file(path, strict=strict)
    out = DeployedWeights(filters=3)
    tables = list(p.tables)
    if tables and len(tables[0]) == 9:
        t = tables.pop(0)
        a = _q(t.rows).reshape(3, 3)            # [f][k0, k1, b]
        out.conv_kernel = np.ascontiguousarray(a[:, :2].T).reshape(1, 2, 1, 3)
        out.conv_bias = a[:, 2].copy()
        out.negzero["conv"] = list(t.negzero_rows)
    if len(p.bare) >= 3:
        out.dense_bias = _q(p.bare[:3])
        out.negzero["dense_bias"] = list(p.bare_negzero)
    dense_tabs = [t for t in tables if len(t) == 387]
    if len(dense_tabs) == 6:
        dk = np.zeros((774, 3), np.float32)
        nz: List[int] = []
        for ti, t in enumerate(dense_tabs):
            c, h = divmod(ti, 2)
            v = _q(t.rows).reshape(3, 129)      # [f][w]
            rows = (h * 387 + np.arange(129)[None, :] * 3 + np.arange(3)[:, None])
            dk[rows, c] = v
            nz += [ti * 387 + r for r in t.negzero_rows]
        out.dense_kernel = dk
        out.negzero["dense"] = nz
        first = dense_tabs[0].rows
        out.placeholder_dense = all(t.rows == first for t in dense_tabs[1:])
    elif dense_tabs:
        raise Q612FormatError(f"{path}: expected 6 dense tables of 387 rows, found {len(dense_tabs)}")
    return out


def load_dense_f10(path: str, *, strict: bool = False) -> DeployedWeights:
    """F=10 dense dump: one flat table of 7740 rows, idx = c*2580 + h*1290 + f*129 + w
    (cnn_test_latest1.sv:711-716).  No conv/bias rows: take those from the .h5."""
    p = parse_file(path, strict=strict)
    if len(p.tables) != 1 or len(p.tables[0]) != 7740:
        raise Q612FormatError(f"{path}: expected one table of 7740 rows")
    t = p.tables[0]
    v = _q(t.rows).reshape(3, 2, 10, 129)       # [c][h][f][w]
    dk = np.ascontiguousarray(v.transpose(1, 3, 2, 0)).reshape(2580, 3)   # rows h*1290 + w*10 + f
    return DeployedWeights(filters=10, dense_kernel=dk, negzero={"dense": list(t.negzero_rows)})


def load_weights_txt(path: str, *, strict: bool = False) -> DeployedWeights:
    p = parse_file(path, strict=strict)
    if len(p.tables) == 1 and len(p.tables[0]) == 7740:
        return load_dense_f10(path, strict=strict)
    return load_weights_f3(path, strict=strict)


def dump_weights_f3(w: DeployedWeights) -> str:
    """Bug-free writer in the layout of 12.15.latestWeights.txt (round-trips with the loader)."""
    lines = ["* Convolution Bias + Weights:", ""]
    k = w.conv_kernel.reshape(2, 3)
    i = 0
    for f in range(3):
        for v in (k[0, f], k[1, f], w.conv_bias[f]):
            lines.append(f"18'd{i:02d}: data <= 18'b{float2fix(float(v))};")
            i += 1
    lines += ["", "", "* Dense Bias:", ""]
    lines += [f"18'b{float2fix(float(v))}" for v in w.dense_bias]
    lines += ["", "", "* Dense Weights (6 Tables):", ""]
    for c in range(3):
        for h in range(2):
            for f in range(3):
                for x in range(129):
                    v = w.dense_kernel[h * 387 + x * 3 + f, c]
                    lines.append(f"18'd{f * 129 + x:03d}: data <= 18'b{float2fix(float(v))};")
            lines.append(" ")
    return "\n".join(lines) + "\n"
