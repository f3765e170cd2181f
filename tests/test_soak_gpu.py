"""Short determinism soak (tools/soak.py): kernels with hand-written synchronisation must return the same bits every time."""
import importlib.util
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_repeated_forwards_are_bit_identical():
    spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    assert soak.run(reps=40, log=lambda *a, **k: None) == []
