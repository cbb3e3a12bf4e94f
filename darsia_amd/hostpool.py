"""Recycled host result arrays.

A fresh ``np.empty`` of a 4000 x 8000 float32 result costs ~8 ms of first-touch page faults when the device-to-host copy
lands in it (10.5 ms against 2.3 ms into pages that are mapped already -- the same 2.3 ms a page-locked buffer gets, measured;
page-locking the recycled buffers, hipHostRegister at 6 ms per 128 MB, was tried and changed nothing in a frame series),
and NumPy hands arrays of that size back to the OS the moment they die.  Frame series (``ConcentrationAnalysis.map``) and
repeated whole-image calls therefore take their result arrays from here: a buffer goes back to the pool when the array the
caller was given -- and every view of it -- has been garbage-collected, and the next result of the same size re-uses its
(mapped) pages.  A caller who keeps every result simply gets fresh arrays, as before.

``DXH_HOST_POOL_MB`` caps what the pool keeps (default 1024; 0 switches it off).  Nothing here touches the device.
"""
from __future__ import annotations

import os
import threading

import numpy as np

_MIN_BYTES = 8 << 20
_lock = threading.Lock()
_free: dict = {}          # nbytes -> [uint8 arrays]
_cached = 0


def _cap() -> int:
    try:
        return int(os.environ.get("DXH_HOST_POOL_MB", "1024")) << 20
    except ValueError:
        return 1024 << 20


class _Holder:
    """Owns one pooled buffer; the array handed out has this object as its base."""

    def __init__(self, buf: np.ndarray):
        self.buf = buf
        self.__array_interface__ = {"data": (buf.ctypes.data, False), "shape": (buf.nbytes,), "typestr": "|u1", "version": 3}

    def __del__(self):
        global _cached
        try:
            buf, n = self.buf, self.buf.nbytes
            with _lock:
                if _cached + n <= _cap():
                    _free.setdefault(n, []).append(buf)
                    _cached += n
        except Exception:   # interpreter shutdown: module globals may be gone
            pass


def empty(shape, dtype) -> np.ndarray:
    """Like ``np.empty(shape, dtype)`` (C order); large arrays come out of the pool when it has a buffer of that size."""
    global _cached
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if n < _MIN_BYTES or _cap() == 0:
        return np.empty(shape, dtype)
    with _lock:
        lst = _free.get(n)
        buf = lst.pop() if lst else None
        if buf is not None:
            _cached -= n
    if buf is None:
        buf = np.empty(n, np.uint8)
    return np.asarray(_Holder(buf)).view(dtype).reshape(shape)


def trim() -> int:
    """Drop every cached buffer; returns the number of bytes released."""
    global _cached
    with _lock:
        n = _cached
        _free.clear()
        _cached = 0
    return n
