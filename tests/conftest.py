import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _dtype_flow_of_the_module(request, monkeypatch):
    """Most modules of this suite pin the NATIVE float32 kernels against the oracle's same-dtype ("strict") flow, bit for bit:
    they run with DXH_PROMOTION=native, the documented switch for that flow.  Modules that pin the product's DEFAULT --
    the reference's own float32 dtype flow, promotion "numpy2" -- say ``DEFAULT_DTYPE_FLOW = True`` and run without it."""
    if getattr(request.module, "DEFAULT_DTYPE_FLOW", False):
        monkeypatch.delenv("DXH_PROMOTION", raising=False)
    else:
        monkeypatch.setenv("DXH_PROMOTION", "native")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "tvd_golden.npz")
    return np.load(path, allow_pickle=False)


def split_last_axis(a):
    """(..., dim) interleaved split variable -> list of planar arrays."""
    return [np.ascontiguousarray(a[..., j]) for j in range(a.shape[-1])]
