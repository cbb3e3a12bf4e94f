"""Multigrid V-cycle solver on the GPU (reference: utils/linear_solvers/mg.py:13-270) for
``mass_coeff * x - diffusion_coeff * laplace(x) = rhs``.

The results are bit-identical to the reference's (which only runs on NumPy < 2: ``np.lib.pad``, mg.py:236),
including its quirks:

* with scalar coefficients the one Jacobi smoother keeps the diagonal it computed at its first call (h = 1)
  for every level and every later ``update_params`` (jacobi.py:101-106);
* ``heterogeneous`` is decided by the constructor arguments (mg.py:72-75); in that mode every V-cycle restricts
  the coefficient arrays and replaces them by their prolongation, so ``mass_coeff`` / ``diffusion_coeff`` of the
  object change with every call (mg.py:124-181) -- mirrored, the attributes are read back after each solve;
* arrays must have exactly ``dim`` axes (mg.py:231-239 pads one pair per spatial axis): ``ValueError`` otherwise;
* with ``tol`` the iterate BEFORE the converged one is returned (mg.py:259-269).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Union

import numpy as np

from ... import _lib
from ...device import DeviceArray, host_empty
from .solver import Solver


class _DeviceSolverHandle:
    """Owner of one ``dxh_solver_t``."""

    def __init__(self, kind, shape, dtype, dim, depth=0, smoother_iterations=1, maxiter=1, tol=None, h=None,
                 heterogeneous=False):
        self.lib = _lib.load()
        _lib.require_gpu()
        if len(shape) != dim:
            raise ValueError(f"arrays must have exactly dim={dim} axes, got shape {tuple(shape)} "
                             "(the reference's solver fails on additional axes)")
        d = _lib.SolverDesc()
        d.struct_size = C.sizeof(_lib.SolverDesc)
        d.device = _lib.default_device()
        d.dtype = _lib.dtype_code(np.dtype(dtype))
        d.dim = dim
        d.n0, d.n1, d.n2 = int(shape[0]), int(shape[1]), int(shape[2]) if dim == 3 else 1
        d.kind = kind
        d.depth, d.smoother_iterations, d.maxiter = int(depth), int(smoother_iterations), int(maxiter)
        d.heterogeneous = int(bool(heterogeneous))
        d.tol = -1.0 if tol is None else float(tol)
        d.h = -1.0 if h is None else float(h)
        d.stream = None
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self._h = C.c_void_p()
        _lib.check(self.lib.dxh_solver_create(C.byref(self._h), C.byref(d)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.dxh_solver_destroy(self._h)
            self._h = None

    __del__ = close

    def _coef(self, value, name):
        if isinstance(value, np.ndarray) and value.ndim > 0:
            if value.shape != self.shape:
                raise ValueError(f"operands could not be broadcast together: {name} has shape {value.shape}, "
                                 f"the arrays {self.shape}")
            return 0.0, np.ascontiguousarray(value, dtype=self.dtype)
        return float(value), None

    def set_params(self, mass, diffusion):
        m_s, m_a = self._coef(mass, "mass_coeff")
        k_s, k_a = self._coef(diffusion, "diffusion_coeff")
        _lib.check(self.lib.dxh_solver_set_params(self._h, m_s, _lib.ptr(m_a), k_s, _lib.ptr(k_a)))

    def solve(self, x0, rhs):
        # device-resident operands (DeviceArray x0 of the solver's dtype) stay in HBM and give a DeviceArray
        on_device = isinstance(x0, DeviceArray)
        for a in (x0, rhs):
            if isinstance(a, DeviceArray) and a.dtype != self.dtype:
                raise TypeError(f"device-resident operands must have the solver's dtype {self.dtype}")
        if not on_device:
            x0 = np.ascontiguousarray(x0, dtype=self.dtype)
        if not isinstance(rhs, DeviceArray):
            rhs = np.ascontiguousarray(rhs, dtype=self.dtype)
        out = DeviceArray(x0.shape, x0.dtype, x0.device) if on_device else host_empty(x0.shape, x0.dtype)
        n = C.c_int32(0)
        _lib.check(self.lib.dxh_solver_solve(self._h, _lib.ptr(x0), _lib.ptr(rhs), _lib.ptr(out), C.byref(n)))
        return out, n.value

    def get_params(self, want_mass, want_diffusion):
        m = host_empty(self.shape, self.dtype) if want_mass else None
        k = host_empty(self.shape, self.dtype) if want_diffusion else None
        _lib.check(self.lib.dxh_solver_get_params(self._h, _lib.ptr(m), _lib.ptr(k)))
        return m, k


class MG(Solver):
    def __init__(self, depth: int = 2, smoother_iterations: int = 5, maxiter: int = 100, tol: Optional[float] = None,
                 dim: int = 2, mass_coeff: Optional[Union[float, np.ndarray]] = None,
                 diffusion_coeff: Optional[Union[float, np.ndarray]] = None, verbose=False) -> None:
        super().__init__(maxiter=maxiter, tol=tol, mass_coeff=mass_coeff, diffusion_coeff=diffusion_coeff, dim=dim,
                         verbose=verbose)
        self.smoother_iterations = smoother_iterations
        self.depth = depth
        self.heterogeneous = isinstance(mass_coeff, np.ndarray) or isinstance(diffusion_coeff, np.ndarray)
        self._dev = None
        self._dirty = True

    def update_params(self, dim=None, mass_coeff=None, diffusion_coeff=None) -> None:
        super().update_params(dim, mass_coeff, diffusion_coeff)
        self._dirty = True

    def _device(self, x0: np.ndarray):
        key = (x0.shape, x0.dtype, self.dim, self.depth, self.smoother_iterations, self.maxiter, self.tol)
        if self._dev is None or self._key != key:
            if self._dev is not None:
                self._dev.close()
            self._dev = _DeviceSolverHandle(_lib.SOLVER_MG, x0.shape, x0.dtype, self.dim, self.depth,
                                            self.smoother_iterations, self.maxiter, self.tol,
                                            heterogeneous=self.heterogeneous)
            self._key = key
            self._dirty = True
        if self._dirty:
            if self.mass_coeff is None or self.diffusion_coeff is None:
                raise ValueError("mass_coeff and diffusion_coeff must be set (update_params)")
            self._dev.set_params(self.mass_coeff, self.diffusion_coeff)
            self._dirty = False
        return self._dev

    def __call__(self, x0: np.ndarray, rhs: np.ndarray) -> np.ndarray:
        if not isinstance(x0, DeviceArray):
            x0 = np.asarray(x0)
        if x0.dtype.kind != "f":
            raise TypeError("MG needs float32 / float64 arrays")
        if tuple(np.shape(rhs) if not isinstance(rhs, DeviceArray) else rhs.shape) != tuple(x0.shape):
            raise ValueError("x0 and rhs must have the same shape")
        dev = self._device(x0)
        out, cycles = dev.solve(x0, rhs)
        if self.heterogeneous:
            m, k = dev.get_params(isinstance(self.mass_coeff, np.ndarray), isinstance(self.diffusion_coeff, np.ndarray))
            if m is not None:
                self.mass_coeff = m
            if k is not None:
                self.diffusion_coeff = k
        if self.verbose:
            print(f"MG: {cycles} of {self.maxiter} V-cycles completed.")
        return out
