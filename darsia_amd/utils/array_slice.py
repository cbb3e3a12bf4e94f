"""Select a range along one axis of an array without knowing the array's rank in advance.

Mirrors the two helpers the reference's numerics use (utils/array_slice.py:13-57): one returns the
index object, the other applies it.  Host-side indexing only -- the device kernels address neighbours
by strides instead.
"""
from __future__ import annotations

from typing import Optional

import numpy as np


def array_slice_argument(a: np.ndarray, axis: int, start: Optional[int], end: Optional[int], step: int = 1) -> tuple:
    """Index tuple that takes ``start:end:step`` on ``axis`` (negative axes wrap) and everything on
    the axes before it; trailing axes are left to NumPy's implicit full slices."""
    index = [slice(None)] * (axis % a.ndim)
    index.append(slice(start, end, step))
    return tuple(index)


def array_slice(a: np.ndarray, axis: int, start: Optional[int], end: Optional[int], step: int = 1) -> np.ndarray:
    """View of ``a`` restricted to ``start:end:step`` along ``axis``."""
    return a[array_slice_argument(a, axis, start, end, step)]
