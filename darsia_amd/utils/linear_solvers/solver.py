"""Solver base class (reference: utils/linear_solvers/solver.py:12-72) for
``mass_coeff * x - diffusion_coeff * laplace(x) = rhs``."""
from __future__ import annotations

import abc
from typing import Optional, Union

import numpy as np


class Solver:
    def __init__(self, maxiter: int = 1, tol: Optional[float] = None, dim: int = 2,
                 mass_coeff: Optional[Union[float, np.ndarray]] = None,
                 diffusion_coeff: Optional[Union[float, np.ndarray]] = None, verbose=False) -> None:
        self.dim = dim
        self.maxiter = maxiter
        self.tol = tol
        self.mass_coeff = mass_coeff
        self.diffusion_coeff = diffusion_coeff
        self.verbose = verbose

    def update_params(self, dim: Optional[int] = None, mass_coeff=None, diffusion_coeff=None) -> None:
        """``None`` keeps the current value (solver.py:40-58)."""
        if dim is not None:
            self.dim = dim
        if mass_coeff is not None:
            self.mass_coeff = mass_coeff
        if diffusion_coeff is not None:
            self.diffusion_coeff = diffusion_coeff

    @abc.abstractmethod
    def __call__(self, x0: np.ndarray, rhs: np.ndarray) -> np.ndarray:
        ...
