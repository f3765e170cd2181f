"""RML2016.10a reader and the data preparation of the reference's cnn.py:42-82 (CNN.ipynb cells 2 and 4).

The dataset (`RML2016.10a_dict.dat` / `.pkl`, DeepSig) is NOT bundled with the reference and cannot be fetched here;
this module is what a user who has the file needs to get from it to `model.predict(X_test)` on the GPU:

    Xd   = load_rml2016(path)                       cnn.py:42-43   cPickle.load(f, encoding="latin1")
    ds   = RML2016(Xd);  ds.mods, ds.snrs           cnn.py:45      sorted key components
    X, lbl = ds.select(mods_chosen, snrs_chosen)    cnn.py:49-59   cell order = mods outer, SNRs inner; np.vstack
    tr, te = split_indices(len(X), 0.5, seed=2016)  cnn.py:66-72   np.random.seed + np.random.choice(replace=False)
    Y_idx = class_indices(lbl, mods_chosen, te)     cnn.py:80-82   mods_chosen.index(lbl[i][0]) (the one-hot's argmax)

The file is a Python-2 pickle of {(modulation: str, snr: int): ndarray (n, 2, 128) float32}.  The reference opens it
with a plain `cPickle.load`, i.e. it runs whatever the file names; this reader does not: a restricted Unpickler admits
the three numpy reconstructors an array pickle needs (`numpy.core.multiarray._reconstruct`, `numpy.ndarray`,
`numpy.dtype`, under their numpy-1 and numpy-2 module names, plus `numpy.core.multiarray.scalar` for numpy-typed SNR
keys and the latin-1 byte-string spelling of Python-3 re-saves) and refuses every other global, object-dtype arrays
included -- the stance of VTCNN2.load_results.
"""
from __future__ import annotations

import os
import pickle
import struct
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

Key = Tuple[str, int]

def _allowed_globals():
    try:
        from numpy._core import multiarray as ma          # numpy >= 2
    except ImportError:                                   # numpy 1.x
        from numpy.core import multiarray as ma
    table = {}
    for mod in ("numpy.core.multiarray", "numpy._core.multiarray"):
        table[(mod, "_reconstruct")] = ma._reconstruct
        table[(mod, "scalar")] = ma.scalar
    table[("numpy", "ndarray")] = np.ndarray
    table[("numpy", "dtype")] = np.dtype
    # a copy of the dataset re-saved from Python 3 at protocol <= 2 spells every bytes payload as
    # _codecs.encode(<str>, "latin1"); admitted in exactly that form (the Python-2 original carries plain str payloads)
    table[("_codecs", "encode")] = _latin1_bytes
    return table


def _latin1_bytes(text, encoding="latin1"):
    if not isinstance(text, str) or str(encoding).lower().replace("-", "").replace("_", "") not in ("latin1", "iso88591"):
        raise pickle.UnpicklingError("only latin-1 encoded byte payloads are admitted")
    return text.encode("latin1")


_DAMAGE = (EOFError, UnicodeError, MemoryError, OverflowError, TypeError, AttributeError, IndexError, KeyError, ImportError,
           RecursionError, SystemError, struct.error)


class _Bounded:
    """The file as the unpickler sees it: a length field that asks for more bytes than the file still holds is refused BEFORE a
    buffer of that size is read into (a flipped bit in an 8-byte length would otherwise be a multi-gigabyte request).  The C
    unpickler's counted-bytes opcodes allocate their destination before they read: such a request either fails at once
    (MemoryError -> UnpicklingError below) or is never touched beyond the bytes the file really has."""

    def __init__(self, fd):
        self.fd = fd
        self.size = os.fstat(fd.fileno()).st_size

    def _check(self, n):
        if n is not None and n >= 0 and n > self.size - self.fd.tell():
            raise pickle.UnpicklingError("pickle data was truncated (a length field runs past the end of the file)")

    def read(self, n=-1):
        self._check(n)
        return self.fd.read(n)

    def readline(self):
        return self.fd.readline()


class _ArraysOnly(pickle.Unpickler):
    """dict / tuple / str / int / float come without globals; arrays need exactly the reconstructors below."""

    _ALLOWED = None

    def find_class(self, module, name):
        if _ArraysOnly._ALLOWED is None:
            _ArraysOnly._ALLOWED = _allowed_globals()
        try:
            return _ArraysOnly._ALLOWED[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError(
                f"an RML2016.10a file holds a dict of numpy arrays only; refusing to resolve {module}.{name}") from None


def load_rml2016(path: str) -> Dict[Key, np.ndarray]:
    """{(modulation, snr): float32 (n, 2, 128)} from the dataset pickle (cnn.py:42-43, without executing the file).
    Raises pickle.UnpicklingError for a file that names any other global or is damaged (truncated, bit-flipped: whatever the
    unpickler trips over comes out as this one type, and no length field is believed beyond the file's own size), ValueError
    for a wrong structure."""
    try:
        with open(path, "rb") as fd:
            obj = _ArraysOnly(_Bounded(fd), encoding="latin1").load()      # Python-2 str payloads -> latin-1, as the reference passes
    except UnicodeError as e:      # (a ValueError by inheritance, but it says "damaged", not "wrong structure")
        raise pickle.UnpicklingError(f"{path}: damaged pickle ({type(e).__name__}: {e})") from e
    except (pickle.UnpicklingError, ValueError):
        raise
    except _DAMAGE as e:      # what a truncated or bit-flipped pickle trips inside the unpickler: one error type for the caller
        raise pickle.UnpicklingError(f"{path}: damaged pickle ({type(e).__name__}: {e})") from e
    if not isinstance(obj, dict) or not obj:
        raise ValueError(f"{path}: expected a non-empty dict keyed by (modulation, snr)")
    out: Dict[Key, np.ndarray] = {}
    for k, v in obj.items():
        if not (isinstance(k, tuple) and len(k) == 2 and isinstance(k[0], str) and isinstance(k[1], (int, np.integer))
                and not isinstance(k[1], bool)):
            raise ValueError(f"{path}: key {k!r} is not (modulation: str, snr: int)")
        if not isinstance(v, np.ndarray) or v.dtype.kind != "f" or v.ndim != 3 or v.shape[1:] != (2, 128):
            raise ValueError(f"{path}: value of {k!r} is not a float array of shape (n, 2, 128)"
                             f" (got {type(v).__name__} {getattr(v, 'dtype', None)} {getattr(v, 'shape', None)})")
        out[(k[0], int(k[1]))] = np.ascontiguousarray(v, dtype=np.float32)
    return out


class RML2016:
    """The dict with the reference's bookkeeping around it."""

    def __init__(self, cells: Dict[Key, np.ndarray]):
        self.cells = cells
        self.mods: List[str] = sorted({m for m, _ in cells})       # cnn.py:45 (j = 0)
        self.snrs: List[int] = sorted({s for _, s in cells})       # cnn.py:45 (j = 1)

    @classmethod
    def load(cls, path: str) -> "RML2016":
        return cls(load_rml2016(path))

    def select(self, mods_chosen: Sequence[str], snrs_chosen: Iterable[int]) -> Tuple[np.ndarray, List[Key]]:
        """X (N, 2, 128) float32 and lbl = [(mod, snr)] per frame, cells stacked modulation-major in the ORDER GIVEN
        (cnn.py:49-59): the class index of a frame is mods_chosen.index(mod), not its rank among the sorted names."""
        snrs_chosen = list(snrs_chosen)
        X, lbl = [], []
        for mod in mods_chosen:
            for snr in snrs_chosen:
                try:
                    frames = self.cells[(mod, int(snr))]
                except KeyError:
                    raise KeyError(f"the dataset has no cell ({mod!r}, {snr}); modulations {self.mods}, SNRs {self.snrs}") from None
                X.append(frames)
                lbl.extend([(mod, int(snr))] * frames.shape[0])
        if not X:
            return np.zeros((0, 2, 128), np.float32), []
        return np.vstack(X), lbl


def split_indices(n_examples: int, train_fraction: float = 0.5, seed: int = 2016) -> Tuple[np.ndarray, np.ndarray]:
    """(train_idx, test_idx) exactly as cnn.py:66-72 draws them (seed 2016, half; CNN.ipynb cell 4: seed 2015, 0.7):
    the legacy global generator seeded with `seed`, `choice(range(n), size=int(n * fraction), replace=False)`, and the
    test indices as the set difference in CPython's set-iteration order -- reproduced with the same constructs so that
    X_test[i] is the reference's X_test[i] (CNN.ipynb cell 18 indexes X_test[2000]); the evaluation's counts do not
    depend on that order."""
    if not 0.0 <= train_fraction <= 1.0:
        raise ValueError("train_fraction must be within [0, 1]")
    n_train = int(n_examples * train_fraction)
    rs = np.random.RandomState(seed)                  # what np.random.seed(seed) re-seeds; same stream, no global state
    train_idx = rs.choice(np.arange(n_examples), size=n_train, replace=False)
    test_idx = np.fromiter(set(range(0, n_examples)) - set(train_idx.tolist()), dtype=np.int64, count=n_examples - n_train)
    return train_idx.astype(np.int64), test_idx


def class_indices(lbl: Sequence[Key], mods_chosen: Sequence[str], idx: Iterable[int]) -> np.ndarray:
    """int32 class index per selected frame: mods_chosen.index(lbl[i][0]) (cnn.py:81-82; what np.argmax recovers from the
    one-hot rows at cnn.py:205)."""
    order = {m: i for i, m in enumerate(mods_chosen)}
    return np.fromiter((order[lbl[int(i)][0]] for i in idx), dtype=np.int32)


def to_onehot(indices: Sequence[int], classes: int = 0) -> np.ndarray:
    """float64 one-hot rows, `max(index) + 1` columns unless `classes` says more (cnn.py:73-79)."""
    a = np.asarray(list(indices), dtype=np.int64)
    width = max(int(classes), int(a.max()) + 1 if a.size else 0)
    out = np.zeros((a.size, width))
    out[np.arange(a.size), a] = 1
    return out


def snrs_of(lbl: Sequence[Key], idx: Iterable[int]) -> np.ndarray:
    """SNR per selected frame (cnn.py:231: test_SNRs)."""
    return np.fromiter((lbl[int(i)][1] for i in idx), dtype=np.int64)
