"""Short determinism soak (tools/soak.py): kernels with hand-written synchronisation must return the same bits every time."""
import importlib.util
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _soak():
    spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    return soak


def test_repeated_forwards_are_bit_identical():
    assert _soak().run(reps=40, log=lambda *a, **k: None) == []


def test_repeated_training_epochs_are_bit_identical():
    assert _soak().run_training(reps=6, log=lambda *a, **k: None) == []
