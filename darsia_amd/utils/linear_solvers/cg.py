"""Conjugate-gradient solver on the GPU (reference: utils/linear_solvers/cg.py:16-59, a wrapper of
``scipy.sparse.linalg.cg`` on the matrix-free operator ``mass_coeff * x - laplace(x, diffusion_coeff=...)``).

The reference passes ``tol=`` to SciPy, which only SciPy < 1.14 accepts; there the iteration stops when
``|r| <= tol * |rhs|`` (legacy ``atol``).  The same rule and the textbook recurrences run here, with the dot
products reduced in double precision; results agree with SciPy's to the solver tolerance, not bit for bit
(SciPy accumulates in BLAS order).  Arrays must have exactly ``dim`` axes (cg.py:58).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from ... import _lib
from ...device import DeviceArray
from .mg import _DeviceSolverHandle
from .solver import Solver


class CG(Solver):
    def __call__(self, x0: np.ndarray, rhs: np.ndarray, h: Optional[float] = None) -> np.ndarray:
        if not isinstance(x0, DeviceArray):     # device-resident operands stay in HBM (DeviceArray in, DeviceArray out)
            x0 = np.asarray(x0)
        if x0.dtype.kind != "f":
            raise TypeError("CG needs float32 / float64 arrays")
        if tuple(rhs.shape if isinstance(rhs, DeviceArray) else np.shape(rhs)) != tuple(x0.shape):
            raise AssertionError("x0 and rhs must have the same shape")  # cg.py:33
        if self.tol is None:
            raise TypeError("CG needs a tolerance (scipy.sparse.linalg.cg(tol=None) fails in the reference as well)")
        if self.mass_coeff is None or self.diffusion_coeff is None:
            raise ValueError("mass_coeff and diffusion_coeff must be set (update_params)")
        dev = _DeviceSolverHandle(_lib.SOLVER_CG, x0.shape, x0.dtype, self.dim, maxiter=self.maxiter, tol=self.tol, h=h)
        try:
            dev.set_params(self.mass_coeff, self.diffusion_coeff)
            out, its = dev.solve(x0, rhs)
        finally:
            dev.close()
        if self.verbose:
            print(f"CG: {its} of {self.maxiter} iterations.")
        return out
