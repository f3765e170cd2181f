import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"          # exists only in the build container, never on the GPU box


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:      # property tests: the same examples on every run (a suite run must not depend on a seed), no wall-clock deadline
        from hypothesis import settings
        settings.register_profile("repo", derandomize=True, deadline=None, database=None)
        settings.load_profile("repo")
    except ImportError:
        pass


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no ROCm device in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def reference_dir():
    if not os.path.isdir(REFERENCE):
        pytest.skip("reference tree not present (GPU box)")
    return REFERENCE


def load_deployed_npz(name):
    z = np.load(os.path.join(GOLDEN, "weights", name + ".npz"))
    return [(z["conv_kernel"], z["conv_bias"]), (z["dense_kernel"], z["dense_bias"])]


H5_NAMES = ["2convmodrecnets_CNN2_0.5", "3convmodrecnets_CNN2_0.5", "4convmodrecnets_CNN2_0.5",
            "5convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5"]
