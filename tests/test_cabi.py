"""The C-ABI library loads and exports every symbol include/mdc.h declares (no GPU needed)."""
import ctypes
import os
import re

from conftest import ROOT
from modulationdetectioncnn_amd import _cabi


def _declared():
    text = open(os.path.join(ROOT, "include", "mdc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdc_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_cabi.EXPORTS)


def test_library_exports_every_declared_symbol():
    import modulationdetectioncnn_amd.build as b
    lib = b.build()
    L = ctypes.CDLL(lib)
    for name in _declared():
        assert hasattr(L, name), name
    assert L.mdc_abi_version() == _cabi.ABI_VERSION == 2


def test_binding_loads_and_reports_errors_without_gpu():
    L = _cabi.lib()
    L.mdc_last_error.restype = ctypes.c_char_p
    # null arguments are rejected before any device call
    assert L.mdc_create(None, 0, None) == -22
    assert b"null" in L.mdc_last_error()
    assert L.mdc_forward(None, None, 0, None, None, None, 0, None, 0, None) == -22
    assert L.mdc_workspace_bytes(None, 10) == 0


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "modulationdetectioncnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle_np" not in src, f
