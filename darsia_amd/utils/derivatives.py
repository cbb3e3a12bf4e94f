"""Finite differences on the GPU (reference: utils/derivatives.py:16-121).

Same names, arguments and results as the reference functions; the arithmetic
runs in ``k_diff`` (darsia_amd/csrc/dxh_kernels.h) in the array's dtype."""
from __future__ import annotations

from typing import Optional, Union

import numpy as np

from .. import _lib


def _diff(img: np.ndarray, axis: int, mode: int, h: Optional[float]) -> np.ndarray:
    if not isinstance(img, np.ndarray):
        raise TypeError("img must be a numpy array")
    code = _lib.dtype_code(img.dtype)
    if img.ndim < 1 or img.ndim > 3:
        raise ValueError("arrays with 1 to 3 axes are supported")
    shape = tuple(img.shape) + (1,) * (3 - img.ndim)
    _lib.require_gpu()
    src = np.ascontiguousarray(img)
    out = np.empty_like(src)
    _lib.check(_lib.load().dxh_diff(_lib.default_device(), code, shape[0], shape[1], shape[2], axis, mode,
                                    -1.0 if h is None else float(h), _lib.ptr(src), _lib.ptr(out)))
    return out


def backward_diff(img: np.ndarray, axis: int, dim: int = 2, h: Optional[float] = None) -> np.ndarray:
    """``img[i+1] - img[i]`` along ``axis``, last entry 0 (derivatives.py:16-37)."""
    assert axis < dim, "axis must be smaller than dimension"
    return _diff(img, axis, 0, h)


def forward_diff(img: np.ndarray, axis: int, dim: int = 2, h: Optional[float] = None) -> np.ndarray:
    """``img[i] - img[i-1]`` along ``axis``, first entry 0 (derivatives.py:40-59)."""
    assert axis < dim, "axis must be smaller than dimension"
    return _diff(img, axis, 1, h)


def laplace(img: np.ndarray, axis: Optional[int] = None, dim: int = 2, h: Optional[float] = None,
            diffusion_coeff: Union[np.ndarray, float] = 1) -> np.ndarray:
    """Laplace operator, composed from the two differences exactly as in derivatives.py:62-121
    (the coefficient product and the 0.5 * sum are single elementwise passes on the host)."""
    if isinstance(diffusion_coeff, np.ndarray):
        assert diffusion_coeff.shape == img.shape

    def one(ax):
        return 0.5 * (
            backward_diff(diffusion_coeff * forward_diff(img, ax, dim, h), ax, dim, h)
            + forward_diff(diffusion_coeff * backward_diff(img, ax, dim, h), ax, dim, h)
        )

    if axis is None:
        res = np.zeros_like(img)
        for ax in range(dim):
            res += one(ax)
        return res
    assert axis < dim, "axis must be smaller than dimension"
    return one(axis)
