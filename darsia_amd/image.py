"""Minimal ``Image`` container: the part of ``darsia.Image`` (image/image.py:35)
the restoration operators touch -- an array plus metadata, ``copy()`` and
``.img`` replacement.  Real ``darsia.Image`` objects work with the operators
as well (duck typing)."""
from __future__ import annotations

import copy as _copy

import numpy as np


class Image:
    def __init__(self, img: np.ndarray, **metadata) -> None:
        if not isinstance(img, np.ndarray):
            raise TypeError("img must be a numpy array")
        self.img = img
        self.metadata_dict = dict(metadata)

    @property
    def shape(self):
        return self.img.shape

    @property
    def dtype(self):
        return self.img.dtype

    def metadata(self) -> dict:
        return _copy.deepcopy(self.metadata_dict)

    def copy(self) -> "Image":
        """Deep copy (image/image.py:363-370)."""
        return _copy.deepcopy(self)
