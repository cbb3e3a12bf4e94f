"""Jacobi solver / smoother on the GPU (reference: utils/linear_solvers/jacobi.py:14-133).

Differences from the reference, both deliberate:

* no hidden ``const_diag`` cache: the diagonal follows ``update_params`` (Q3);
* two dtype flows for float32 operands, as in the TV loop (``promotion``, ``DXH_PROMOTION``): by default the
  reference's own under NumPy >= 2 -- with Python-float coefficients its ``const_diag_scaled`` is an
  ``np.float64``, so ``rhs / diag`` stays float32, the first sweep sums the neighbours of the float32 ``x0`` in
  float32, and from its division on the iterate is float64: ``Jacobi.__call__`` RETURNS A FLOAT64 ARRAY for
  float32 input there, and so does this one (``dxh_jacobi_solve_np2``); ``"native"`` keeps everything float32.
  Array-valued coefficients always take the native flow.

Kept: the neighbour sum runs over EVERY array axis (Q2), and with ``tol`` the
iterate BEFORE the one that met the tolerance is returned (Q7).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ... import _lib
from ...device import DeviceArray, host_empty
from .solver import Solver


def _coef(value, like: np.ndarray):
    """(scalar, array-or-None) view of a coefficient."""
    if isinstance(value, np.ndarray) and value.ndim > 0:
        arr = np.ascontiguousarray(np.broadcast_to(value, like.shape), dtype=like.dtype)
        return 0.0, arr
    return float(value), None


def float32_flow(promotion, *coefficients) -> str:
    """The dtype flow of a float32 solve: ``promotion`` if given, else ``DXH_PROMOTION``, else the reference's own
    ("numpy2") -- which exists for scalar coefficients only."""
    import os

    if promotion is None:
        promotion = os.environ.get("DXH_PROMOTION") or "numpy2"
    if promotion not in ("native", "numpy2"):
        raise ValueError(f"promotion must be None, 'native' or 'numpy2', not {promotion!r}")
    if any(np.ndim(c) != 0 for c in coefficients):
        return "native"
    return promotion



def _verbose_lines(done: int, maxiter: int, tol) -> None:
    """What the reference prints (jacobi.py:115,127-131): one line per sweep.  Without a tolerance the lines carry no numbers and
    are reproduced as they are; with one the reference also prints each sweep's increment, which stays on the device here -- a
    summary line instead."""
    if tol is None:
        for i in range(done):
            print(f"Jacobi iteration {i} of {maxiter} completed.")
    else:
        print(f"Jacobi: {done} of {maxiter} sweeps completed.")

class Jacobi(Solver):
    def __call__(self, x0: np.ndarray, rhs: np.ndarray, h: float = 1.0, promotion=None) -> np.ndarray:
        if self.mass_coeff is None or self.diffusion_coeff is None:
            raise ValueError("mass_coeff and diffusion_coeff must be set (update_params)")
        # device-resident operands stay in HBM: DeviceArray in (x0, and rhs of the same dtype), DeviceArray out
        on_device = isinstance(x0, DeviceArray)
        if not on_device:
            x0 = np.asarray(x0)         # lists and other array-likes, as the reference's NumPy expressions take them
        if not isinstance(rhs, DeviceArray):
            rhs = np.asarray(rhs)
        # the reference's flow under NumPy >= 2 as reproduced here hinges on PYTHON-number coefficients (weak scalars; only
        # their quotient const_diag_scaled is an np.float64); NumPy scalars (np.float32(..), np.float64(..)) are strongly
        # typed and promote differently at other places -- they stay with the native flow
        weak = all(type(c) in (float, int) for c in (self.mass_coeff, self.diffusion_coeff))
        if (x0.dtype == np.float32 and rhs.dtype == np.float32 and weak
                and float32_flow(promotion, self.mass_coeff, self.diffusion_coeff) == "numpy2"):
            return self._call_numpy2(x0, rhs, h)
        code = _lib.dtype_code(x0.dtype)
        if isinstance(rhs, DeviceArray):
            if rhs.dtype != x0.dtype:
                raise TypeError("a device-resident rhs must have x0's dtype")
        else:
            rhs = np.ascontiguousarray(rhs, dtype=x0.dtype)
        if tuple(rhs.shape) != tuple(x0.shape):
            raise ValueError("x0 and rhs must have the same shape")
        ndim, n0, n1, n2 = _lib.shape3(x0.shape, min(self.dim, x0.ndim))
        _lib.require_gpu()
        like = np.empty((), x0.dtype) if on_device else x0
        m_s, m_a = _coef(self.mass_coeff, np.broadcast_to(like, x0.shape) if on_device else x0)
        k_s, k_a = _coef(self.diffusion_coeff, np.broadcast_to(like, x0.shape) if on_device else x0)
        src = x0 if on_device else np.ascontiguousarray(x0)
        out = DeviceArray(src.shape, src.dtype, src.device) if on_device else host_empty(src.shape, src.dtype)
        done = C.c_int(0)
        tol = -1.0 if self.tol is None else float(self.tol)
        _lib.check(_lib.load().dxh_jacobi_solve(
            src.device if on_device else _lib.default_device(), code, ndim, self.dim, _lib.NSUM_ALL_AXES, n0, n1, n2,
            _lib.ptr(src), _lib.ptr(rhs), _lib.ptr(out), m_s, _lib.ptr(m_a), k_s, _lib.ptr(k_a), float(h),
            int(self.maxiter), tol, C.byref(done)))
        if self.verbose:
            _verbose_lines(done.value, self.maxiter, self.tol)
        return out

    def _call_numpy2(self, x0, rhs, h: float, out_dtype=np.float64, nsum=_lib.NSUM_ALL_AXES):
        """float32 operands in the reference's dtype flow (``dxh_jacobi_solve_np2``); ``rhs = None``: ``mass * x0``."""
        on_device = isinstance(x0, DeviceArray)
        src = x0 if on_device else np.ascontiguousarray(x0)
        if rhs is not None:
            if tuple(rhs.shape) != tuple(x0.shape):
                raise ValueError("x0 and rhs must have the same shape")
            if not isinstance(rhs, DeviceArray):
                rhs = np.ascontiguousarray(rhs, dtype=np.float32)
        ndim, n0, n1, n2 = _lib.shape3(src.shape, min(self.dim, src.ndim))
        _lib.require_gpu()
        out_dtype = np.dtype(out_dtype)
        out = DeviceArray(src.shape, out_dtype, src.device) if on_device else host_empty(src.shape, out_dtype)
        done = C.c_int(0)
        _lib.check(_lib.load().dxh_jacobi_solve_np2(
            src.device if on_device else _lib.default_device(), ndim, self.dim, nsum, n0, n1, n2, _lib.ptr(src), _lib.ptr(rhs),
            _lib.ptr(out), _lib.dtype_code(out_dtype), float(self.mass_coeff), float(self.diffusion_coeff), float(h),
            int(self.maxiter), -1.0 if self.tol is None else float(self.tol), C.byref(done)))
        if self.verbose:
            _verbose_lines(done.value, self.maxiter, self.tol)
        if done.value == 0 and out_dtype != np.float32:
            # no sweep was accepted (a tolerance met by the very first one, jacobi.py:123-126): the reference returns x0
            # itself, a float32 array
            if on_device:
                out.close()
                return src
            return np.array(src, dtype=np.float32)
        return out
