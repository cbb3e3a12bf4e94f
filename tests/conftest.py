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
    """The suite runs in the flow a user of the drop-in API gets: the product's DEFAULT (the reference's own float32 dtype
    flow under NumPy >= 2, promotion "numpy2", wherever it exists) -- no DXH_PROMOTION in the environment.  Modules that pin
    the NATIVE float32 kernels (the reference's arithmetic under NumPy < 2, what bench.py's headline times) against the
    oracle's same-dtype ("strict") flow or the NumPy-1.26 goldens say ``NATIVE_DTYPE_FLOW = True`` and run with
    DXH_PROMOTION=native, the documented switch for that flow."""
    if getattr(request.module, "NATIVE_DTYPE_FLOW", False):
        monkeypatch.setenv("DXH_PROMOTION", "native")
    else:
        monkeypatch.delenv("DXH_PROMOTION", raising=False)


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "tvd_golden.npz")
    return np.load(path, allow_pickle=False)


def split_last_axis(a):
    """(..., dim) interleaved split variable -> list of planar arrays."""
    return [np.ascontiguousarray(a[..., j]) for j in range(a.shape[-1])]
