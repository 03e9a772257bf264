"""pytest configuration: markers, repo root on sys.path, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def oracle():
    """TEST-ONLY CPU oracle (restatement of the reference qr.c)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def qr():
    """The product binding (ctypes over the C-ABI library).  Fails loudly if the .so is missing."""
    import cuda_qr_amd
    return cuda_qr_amd
