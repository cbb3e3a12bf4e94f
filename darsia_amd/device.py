"""Device-resident arrays: chaining operators without leaving HBM.

The reference chains its operators in host memory (``Image.__init__`` applies the
corrections in place, image/image.py:188-199; ``ColorCorrection.correct_array``
hands its float32 result to whatever comes next, corrections/color/colorcorrection.py:388-394;
``CombinedModel`` calls ``Resize -> TVD -> Resize`` one after the other, signals/models/combinedmodel.py:29-52).
Here every hop through a NumPy array would be a trip over PCIe (7-8 ms per 384 MB
frame each way, several times a kernel's run time).  A ``DeviceArray`` is a dense
C-contiguous array in HBM; an operator that is given one returns one (same type
out as in, like the reference's ndarray | Image protocol), and the C ABI takes its
pointer wherever it takes a host array's (every copy of caller data resolves its
direction from the pointers).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .hostpool import empty as host_empty

from . import _lib


class DeviceArray:
    """A dense, C-contiguous array in device memory (``dxh_mem_alloc``)."""

    def __init__(self, shape, dtype, device: Optional[int] = None):
        self.lib = _lib.load()
        _lib.require_gpu()
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.device = _lib.default_device() if device is None else int(device)
        p = C.c_void_p()
        _lib.check(self.lib.dxh_mem_alloc(self.device, max(1, self.nbytes), C.byref(p)))
        self.ptr = p.value

    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def size(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1

    @property
    def nbytes(self) -> int:
        return self.size * self.dtype.itemsize

    @classmethod
    def from_host(cls, arr: np.ndarray, device: Optional[int] = None) -> "DeviceArray":
        arr = np.ascontiguousarray(arr)
        d = cls(arr.shape, arr.dtype, device)
        if arr.nbytes:
            _lib.check(d.lib.dxh_mem_copy(d.device, d.ptr, _lib.ptr(arr), arr.nbytes))
        return d

    def get(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Copy to the host (into ``out`` if given: same shape and dtype, C-contiguous)."""
        if out is None:
            out = host_empty(self.shape, self.dtype)
        assert out.shape == self.shape and out.dtype == self.dtype and out.flags.c_contiguous
        if self.nbytes:
            _lib.check(self.lib.dxh_mem_copy(self.device, _lib.ptr(out), self.ptr, self.nbytes))
        return out

    def set(self, arr: np.ndarray) -> "DeviceArray":
        """Overwrite with a host array of the same shape and dtype."""
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert tuple(arr.shape) == self.shape
        if self.nbytes:
            _lib.check(self.lib.dxh_mem_copy(self.device, self.ptr, _lib.ptr(arr), self.nbytes))
        return self

    def copy(self) -> "DeviceArray":
        """A device-to-device copy (what ``CombinedModel`` starts from, signals/models/combinedmodel.py:39)."""
        d = DeviceArray(self.shape, self.dtype, self.device)
        if self.nbytes:
            _lib.check(self.lib.dxh_mem_copy(self.device, d.ptr, self.ptr, self.nbytes))
        return d

    def __deepcopy__(self, memo) -> "DeviceArray":
        return self.copy()   # Image.copy() is copy.deepcopy: the pixels are copied inside HBM

    def reshape(self, *shape) -> "DeviceArray":
        """A view with another shape (shares the memory; keep the original alive)."""
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if int(np.prod(shape, dtype=np.int64)) != self.size:
            raise ValueError(f"cannot reshape {self.shape} into {tuple(shape)}")
        v = object.__new__(DeviceArray)
        v.lib, v.shape, v.dtype, v.device, v.ptr, v._base = self.lib, tuple(int(s) for s in shape), self.dtype, self.device, self.ptr, self
        return v

    def close(self):
        if getattr(self, "ptr", None) and not hasattr(self, "_base"):
            self.lib.dxh_mem_free(self.device, self.ptr)
        self.ptr = None

    __del__ = close

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, device={self.device})"


def is_device_array(x) -> bool:
    return isinstance(x, DeviceArray)


def trim_device_cache() -> int:
    """Hand the device planes the TV contexts keep for re-use back to the HIP runtime (``dxh_mem_trim``); returns the
    number of bytes released."""
    import ctypes as C

    n = C.c_size_t(0)
    _lib.check(_lib.load().dxh_mem_trim(C.byref(n)))
    return n.value
