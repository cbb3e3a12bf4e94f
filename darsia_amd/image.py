"""``Image`` container: the part of ``darsia.Image`` (image/image.py:35-546) the restoration,
registration and mask operators touch -- pixel array plus spatial metadata (space dimension,
matrix indexing, physical dimensions, origin), ``copy()``, ``metadata()`` and a coordinate system.
No I/O, plotting, time series or arithmetic.  Real ``darsia.Image`` objects work with the operators
as well (duck typing on ``.img`` / ``.copy()``)."""
from __future__ import annotations

import copy as _copy

import numpy as np

from .geometry import Coordinate, CoordinateSystem, interpret_indexing


class Image:
    def __init__(self, img: np.ndarray, transformations=None, **kwargs) -> None:
        if not isinstance(img, np.ndarray):
            raise TypeError("img must be a numpy array")
        self.img = img
        self.original_dtype = img.dtype
        self.space_dim: int = kwargs.pop("space_dim", 2)
        self.indexing: str = kwargs.pop("indexing", "ijk"[: self.space_dim])
        assert self.indexing == "ijk"[: self.space_dim]
        self.dimensions = list(kwargs.pop("dimensions", self.space_dim * [1]))
        for key, pos in (("height", 0), ("width", 1), ("depth", 2)):
            if key in kwargs:
                self.dimensions[pos] = kwargs.pop(key)
        # default origin: the Cartesian position of voxel (0, 0[, 0]); reverted axes start at their extent
        default_origin = self.space_dim * [0]
        for counter, index in enumerate(self.indexing):
            axis, reverse = interpret_indexing(index, "xyz"[: self.space_dim])
            if reverse:
                default_origin[axis] = self.dimensions[counter]
        self.origin = Coordinate(np.array(kwargs.pop("origin", default_origin)))
        self.name = kwargs.pop("name", None)
        self.scalar = kwargs.pop("scalar", False)
        self.series = False
        self.time_dim = 0
        self.extra_metadata = dict(kwargs)  # user tags travel with the image
        # corrections given at construction are applied in place, in order (image/image.py:188-199)
        if transformations is not None:
            for transformation in transformations:
                if transformation is not None and hasattr(transformation, "__call__"):
                    transformation(self, overwrite=True)

    @property
    def shape(self):
        return self.img.shape

    @property
    def dtype(self):
        return self.img.dtype

    @property
    def num_voxels(self) -> list:
        return list(self.shape[: self.space_dim])

    @property
    def voxel_size(self) -> list:
        return [self.dimensions[i] / self.num_voxels[i] for i in range(self.space_dim)]

    @property
    def coordinatesystem(self) -> CoordinateSystem:
        return CoordinateSystem(self)

    def metadata(self) -> dict:
        """Keyword arguments that rebuild the spatial description (image/image.py:525-546)."""
        meta = {
            "space_dim": self.space_dim,
            "indexing": self.indexing,
            "dimensions": list(self.dimensions),
            "origin": np.array(self.origin),
            "name": self.name,
            "scalar": self.scalar,
        }
        meta.update(_copy.deepcopy(self.extra_metadata))
        return meta

    def update_metadata(self, meta=None, **kwargs) -> None:
        """Set attributes from ``meta`` (keys of ``metadata()``) and keyword arguments (image/image.py:441-454)."""
        for key, value in {**(meta or {}), **kwargs}.items():
            setattr(self, key, value)

    def copy(self) -> "Image":
        """Deep copy (image/image.py:363-370)."""
        return _copy.deepcopy(self)


class ScalarImage(Image):
    """Single-channel image (image/image.py:1860)."""

    def __init__(self, img: np.ndarray, transformations=None, **kwargs) -> None:
        kwargs["scalar"] = True
        super().__init__(img, transformations, **kwargs)


class ExtensiveImage(ScalarImage):
    """Extensive quantity stored per voxel (image/image.py:2064-2078): resizing it conserves the sum by default
    (restoration/resize.py:135-190)."""

    extensive = True


class OpticalImage(Image):
    """Photograph with a trailing colour axis (image/image.py:2081)."""

    def __init__(self, img: np.ndarray, transformations=None, **kwargs) -> None:
        kwargs["scalar"] = False
        super().__init__(img, transformations, **kwargs)


def with_array(img, array: np.ndarray):
    """A copy of the Image-like ``img`` (any object with ``.img`` and ``.copy()``, e.g. ``darsia.Image``) that carries
    ``array`` instead of its pixels.  Equivalent to ``out = img.copy(); out.img = array`` as the reference writes it
    (tvd.py:127-129, piecewiseperspective.py:143-145; ``Image.copy`` is ``copy.deepcopy``, image/image.py:363-370)
    without deep-copying the old pixel array first -- 40 ms per 384 MB.  The input object is never modified (other
    threads may be reading it): the deep copy runs with the old pixel array pre-mapped to ``array`` in its memo, so the
    copy machinery substitutes instead of copying."""
    pixels = getattr(img, "img", None)
    try:
        out = _copy.deepcopy(img, {id(pixels): array})
    except Exception:  # noqa: BLE001 -- containers that cannot be deep-copied generically: their own copy()
        out = None
    if out is None or getattr(out, "img", None) is not array:
        out = img.copy()
        out.img = array
    return out
