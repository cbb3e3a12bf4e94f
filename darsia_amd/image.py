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
        from .device import DeviceArray

        if not isinstance(img, (np.ndarray, DeviceArray)):   # a DeviceArray: pixels that stay in HBM between device operators
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

    # ---- type changes (image/image.py:461-521) ---------------------------------------------------------------
    def astype(self, data_type):
        """A copy with the pixel array cast to a scalar type, or re-wrapped as another Image class."""
        if data_type in (int, float, np.uint8, np.uint16, np.float16, np.float32, np.float64, bool):
            out = self.copy()
            out.img = out.img.astype(data_type)
            return out
        return data_type(img=self.img.copy(), **self.metadata())

    def img_as(self, data_type):
        """A copy with the pixel array converted the scikit-image way (range-aware: uint8 <-> [0, 1] floats ...);
        the conversions of this path -- float, float32, float64, uint8, uint16 -- follow ``darsia_amd.utils.dtype``."""
        from .utils.dtype import convert_dtype

        out = self.copy()
        if data_type in (float, np.float64):
            out.img = convert_dtype(out.img, np.float64) if out.img.dtype.kind != "f" or data_type is np.float64 else out.img
        elif data_type is np.float32:
            out.img = convert_dtype(out.img, np.float32)
        elif data_type is np.uint8:
            out.img = convert_dtype(np.clip(out.img, -1, 1) if out.img.dtype.kind == "f" else out.img, np.uint8)
        elif data_type is np.uint16:
            out.img = convert_dtype(out.img, np.uint16)
        else:
            raise NotImplementedError
        return out

    # ---- extraction (image/image.py:665-746) -----------------------------------------------------------------
    def subregion(self, roi):
        """The image restricted to a box: a tuple of slices (voxels), a ``VoxelArray`` or a ``CoordinateArray`` of points
        spanning the box; dimensions and origin of the result describe the cut-out."""
        from .geometry import CoordinateArray, VoxelArray, interpret_indexing as _interp

        if isinstance(roi, CoordinateArray):
            box = self.coordinatesystem.voxel(roi)
        elif isinstance(roi, VoxelArray):
            box = roi
        elif isinstance(roi, tuple):
            box = None
        else:
            raise ValueError(f"roi is of type {type(roi)}, but needs to be tuple of slices, VoxelArray, or CoordinateArray.")
        voxels = roi if box is None else tuple(
            slice(max(0, int(np.min(box[:, d]))), min(int(np.max(box[:, d])), self.num_voxels[d])) for d in range(self.space_dim))
        assert len(voxels) == self.space_dim
        cs = self.coordinatesystem
        origin = cs.coordinate([0 if sl.start is None else sl.start for sl in voxels])
        opposite = cs.coordinate([self.num_voxels[i] if sl.stop is None else sl.stop for i, sl in enumerate(voxels)])
        extent = np.absolute(np.asarray(opposite) - np.asarray(origin))
        dimensions = [extent[_interp("ijk"[k], "xyz"[: self.space_dim])[0]] for k in range(self.space_dim)]
        meta = self.metadata()
        meta["dimensions"] = dimensions
        meta["origin"] = origin
        return type(self)(self.img[voxels], **meta)

    # ---- arithmetic and comparisons (image/image.py:927-1058) -----------------------------------------------
    def __add__(self, other):
        if self.img.shape != other.img.shape:
            raise ValueError("Images have different shapes.")
        return type(self)(self.img + other.img, **self.metadata())

    def __sub__(self, other):
        if self.img.shape != other.img.shape:
            raise ValueError("Images have different shapes.")
        return type(self)(self.img - other.img, **self.metadata())

    def __mul__(self, weight):
        if not isinstance(weight, (float, int, np.ndarray)):
            raise ValueError
        return type(self)(self.img * weight, **self.metadata())

    __rmul__ = __mul__

    def _compare(self, other, op):
        result = ScalarImage(np.zeros(self.num_voxels, dtype=bool), **{k: v for k, v in self.metadata().items() if k != "scalar"})
        result.img = op(self.img, other.img if isinstance(other, Image) else other)
        return result

    def __lt__(self, other):
        return self._compare(other, np.less)

    def __gt__(self, other):
        return self._compare(other, np.greater)

    def __le__(self, other):
        return self._compare(other, np.less_equal)

    def __ge__(self, other):
        return self._compare(other, np.greater_equal)

    def __eq__(self, other):
        return self._compare(other, np.equal)

    __hash__ = object.__hash__   # comparisons are elementwise; identity hashing keeps images usable as dict keys


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
