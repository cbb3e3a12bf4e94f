import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "tvd_golden.npz")
    return np.load(path, allow_pickle=False)


def split_last_axis(a):
    """(..., dim) interleaved split variable -> list of planar arrays."""
    return [np.ascontiguousarray(a[..., j]) for j in range(a.shape[-1])]
