"""Axis slicing helpers (reference: utils/array_slice.py:13-57)."""
from __future__ import annotations

from typing import Optional

import numpy as np


def array_slice_argument(a: np.ndarray, axis: int, start: Optional[int], end: Optional[int], step: int = 1):
    """Index tuple selecting ``start:end:step`` along ``axis``."""
    return (slice(None),) * (axis % a.ndim) + (slice(start, end, step),)


def array_slice(a: np.ndarray, axis: int, start: Optional[int], end: Optional[int], step: int = 1) -> np.ndarray:
    return a[array_slice_argument(a, axis, start, end, step)]
