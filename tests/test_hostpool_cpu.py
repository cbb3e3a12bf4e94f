"""The pool of host result arrays (darsia_amd/hostpool.py): a buffer is re-used only once nothing refers to it."""
import gc

import numpy as np
import pytest

from darsia_amd import hostpool

BIG = (2048, 2048)   # 16 MB of float32: above the pool's threshold


@pytest.fixture(autouse=True)
def _clean(monkeypatch):
    monkeypatch.delenv("DXH_HOST_POOL_MB", raising=False)
    hostpool.trim()
    yield
    hostpool.trim()


def test_small_arrays_are_plain_numpy_arrays():
    a = hostpool.empty((4, 5), np.float64)
    assert a.base is None and a.shape == (4, 5) and a.dtype == np.float64


def test_a_dead_result_is_recycled_and_a_live_one_never_is():
    a = hostpool.empty(BIG, np.float32)
    assert a.shape == BIG and a.dtype == np.float32 and a.flags.c_contiguous and a.flags.writeable
    a[...] = 3.0
    p = a.ctypes.data
    b = hostpool.empty(BIG, np.float32)
    assert b.ctypes.data != p          # a is alive
    view = a[100:200, ::2]
    del a
    gc.collect()
    c = hostpool.empty(BIG, np.float32)
    assert c.ctypes.data != p          # the view keeps the buffer out of the pool
    assert float(view[0, 0]) == 3.0
    del view
    gc.collect()
    d = hostpool.empty(BIG, np.float32)
    assert d.ctypes.data == p          # now it comes back
    e = hostpool.empty((BIG[0] // 2, BIG[1]), np.float64)   # same byte count, another dtype / shape
    assert e.nbytes == d.nbytes and e.ctypes.data not in (d.ctypes.data, b.ctypes.data, c.ctypes.data)


def test_cap_and_switch(monkeypatch):
    monkeypatch.setenv("DXH_HOST_POOL_MB", "20")   # room for one 16 MB buffer
    a, b = hostpool.empty(BIG, np.float32), hostpool.empty(BIG, np.float32)
    del a, b
    gc.collect()
    assert hostpool.trim() == 16 << 20
    monkeypatch.setenv("DXH_HOST_POOL_MB", "0")
    a = hostpool.empty(BIG, np.float32)
    assert a.base is None
    del a
    gc.collect()
    assert hostpool.trim() == 0


def test_results_survive_pickling_and_copying():
    import copy
    import pickle

    a = hostpool.empty(BIG, np.float32)
    a[...] = 2.0
    b = pickle.loads(pickle.dumps(a))
    c = copy.deepcopy(a)
    assert b.base is None or b.ctypes.data != a.ctypes.data
    assert np.array_equal(a, b) and np.array_equal(a, c)
