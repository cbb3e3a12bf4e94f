"""Spatial metadata for images: points, axis conventions, coordinate system, ROI masks.

The subset of DarSIA's data model the hot path and its callers touch
(reference: utils/point.py:14-260, image/indexing.py:29-160,
image/coordinatesystem.py:11-265, utils/standard_images.py:14-137).  This is
host-side index arithmetic (no pixel data is processed here); masks are
integer/boolean work and must be bit-exact, which ``tests/test_geometry.py``
checks against hand-computed index ranges and the properties the reference's
own ``tests/unit/test_roi_to_mask.py`` states.
"""
from __future__ import annotations

from typing import Union

import numpy as np


# ---- points (utils/point.py) ----------------------------------------------------------------------
class _Point(np.ndarray):
    def __new__(cls, input_array=None):
        if input_array is None:
            input_array = np.empty((0,))
        return np.asarray(input_array).view(cls)


class Coordinate(_Point):
    """Cartesian coordinate (x, y[, z])."""


class Voxel(_Point):
    """Voxel in matrix indexing (row, col[, depth]); integer valued."""

    def __new__(cls, input_array, matrix_indexing: bool = True):
        obj = np.asarray(input_array).astype(int).view(cls)
        if not matrix_indexing:
            obj = np.fliplr(np.atleast_2d(obj)).reshape(obj.shape).view(cls)
        return obj


class CoordinateArray(Coordinate):
    def __getitem__(self, key):
        if isinstance(key, int):
            return Coordinate(np.asarray(self)[key])
        if isinstance(key, np.ndarray) and key.ndim == 1:
            return CoordinateArray(np.asarray(self)[key])
        return np.asarray(self)[key]


class VoxelArray(Voxel):
    def __getitem__(self, key):
        if isinstance(key, int):
            return Voxel(np.asarray(self)[key])
        if isinstance(key, np.ndarray) and key.ndim == 1:
            return VoxelArray(np.asarray(self)[key])
        return np.asarray(self)[key]


def make_coordinate(pts) -> Union[Coordinate, CoordinateArray]:
    pts = np.array(pts)
    if pts.ndim == 1:
        return Coordinate(pts)
    assert pts.ndim == 2 and pts.shape[1] in (1, 2, 3), f"only support 1d, 2d, 3d, received {pts.shape}"
    return CoordinateArray(pts)


def make_voxel(pts, matrix_indexing: bool = True) -> Union[Voxel, VoxelArray]:
    arr = np.array(pts)
    if arr.ndim == 1:
        return Voxel(arr, matrix_indexing=matrix_indexing)
    assert arr.ndim == 2 and arr.shape[1] in (1, 2, 3), f"only support 1d, 2d, 3d, received {arr.shape}"
    return VoxelArray(arr, matrix_indexing=matrix_indexing)


# ---- axis conventions (image/indexing.py) ---------------------------------------------------------
_INTERPRET = {
    "x": {"x": (0, False), "i": (0, False)},
    "i": {"x": (0, False), "i": (0, False)},
    "xy": {"x": (0, False), "y": (1, False), "i": (1, True), "j": (0, False)},
    "ij": {"x": (1, False), "y": (0, True), "i": (0, False), "j": (1, False)},
    "xyz": {"x": (0, False), "y": (1, False), "z": (2, False), "i": (2, True), "j": (0, False), "k": (1, True)},
    "ijk": {"x": (1, False), "y": (2, True), "z": (0, True), "i": (0, False), "j": (1, False), "k": (2, False)},
}


def interpret_indexing(axis: str, indexing: str) -> tuple:
    """(component of ``axis`` in an array indexed by ``indexing``, whether the axis is reverted)."""
    try:
        return _INTERPRET[indexing][axis]
    except KeyError:
        raise ValueError from None


def to_cartesian_indexing(axis, indexing: str) -> str:
    if isinstance(axis, int):
        axis = "ijk"[axis]
    table = {"ij": {"i": "y", "j": "x"}, "ijk": {"i": "z", "j": "y", "k": "x"}}
    try:
        return table[indexing][axis]
    except KeyError:
        raise ValueError from None


def to_matrix_indexing(axis, indexing: str) -> str:
    if isinstance(axis, int):
        axis = "xyz"[axis]
    table = {"xy": {"x": "j", "y": "i"}, "xyz": {"x": "k", "y": "j", "z": "i"}}
    try:
        return table[indexing][axis]
    except KeyError:
        raise ValueError from None


# ---- coordinate system (image/coordinatesystem.py) ------------------------------------------------
class CoordinateSystem:
    """Voxel <-> Cartesian conversion of an image with matrix indexing ("ij" / "ijk")."""

    def __init__(self, img) -> None:
        assert img.indexing in ("i", "ij", "ijk"), "Indexing not supported."
        self.indexing = img.indexing
        self.dim = img.space_dim
        self.shape = img.img.shape[: self.dim]
        self.dimensions = img.dimensions
        self.axes = "xyz"[: self.dim]
        self.voxel_size = {}
        for axis in self.axes:
            pos, _ = interpret_indexing(axis, self.indexing)
            self.voxel_size[axis] = img.voxel_size[pos]
        self._coordinate_of_origin_voxel = img.origin
        self._coordinate_of_opposite_voxel = self.coordinate(img.img.shape[: self.dim])
        corners = np.vstack((self._coordinate_of_origin_voxel, self._coordinate_of_opposite_voxel))
        self.domain = {}
        for i, axis in enumerate(self.axes):
            self.domain[axis + "min"] = np.min(corners[:, i])
            self.domain[axis + "max"] = np.max(corners[:, i])
        self.max_coordinate = np.max(corners, axis=0)
        self.min_coordinate = np.min(corners, axis=0)

    @property
    def voxels(self):
        """All voxels of the image, first index fastest (image/coordinatesystem.py:89-101)."""
        if not hasattr(self, "_voxels"):
            self._voxels = make_voxel(np.indices(self.shape, dtype=int).reshape((self.dim, -1), order="F").T)
        return self._voxels

    @property
    def coordinates(self):
        """Coordinates of all voxels (image/coordinatesystem.py:103-113)."""
        if not hasattr(self, "_coordinates"):
            self._coordinates = self.coordinate(self.voxels)
        return self._coordinates

    def length(self, num, axis: str):
        assert axis in self.axes
        return num * self.voxel_size[axis]

    def num_voxels(self, length, axis: str):
        """Voxels touched by ``length`` (ceil)."""
        assert axis in self.axes
        return np.ceil(length / self.voxel_size[axis]).astype(int)

    def coordinate(self, voxel):
        """(row, col) -> (x, y) incl. scaling; one voxel per row."""
        if isinstance(voxel, tuple):
            voxel = list(voxel)
        if isinstance(voxel, list):
            voxel = np.array(voxel)
        assert isinstance(voxel, np.ndarray)
        varr = np.atleast_2d(voxel)
        coord = np.empty_like(varr, dtype=float)
        for i, axis in enumerate(self.axes):
            pos, revert = interpret_indexing(axis, self.indexing)
            sign = -1 if revert else 1
            coord[:, i] = self._coordinate_of_origin_voxel[i] + sign * varr[:, pos] * self.voxel_size[axis]
        return make_coordinate(coord.reshape(voxel.shape))

    def voxel(self, coordinate):
        """(x, y) -> (row, col): floor((x - origin) / voxel size) per axis, one coordinate per row."""
        if isinstance(coordinate, list):
            coordinate = np.array(coordinate)
        assert isinstance(coordinate, np.ndarray)
        carr = np.atleast_2d(coordinate)
        pixel = np.empty_like(carr, dtype=int)
        for i, axis in enumerate(self.axes):
            pos, revert = interpret_indexing(axis, self.indexing)
            sign = -1 if revert else 1
            pixel[:, pos] = np.floor(sign * (carr[:, i] - self._coordinate_of_origin_voxel[i]) / self.voxel_size[axis])
        return make_voxel(np.round(pixel.reshape(coordinate.shape)).astype(int))

    def coordinate_vector(self, pixel_vector: np.ndarray) -> np.ndarray:
        """Relative displacement in voxels -> metric vector."""
        parr = np.atleast_2d(pixel_vector)
        out = np.empty_like(parr, dtype=float)
        for i, axis in enumerate(self.axes):
            pos, revert = interpret_indexing(axis, self.indexing)
            sign = -1 if revert else 1
            out[:, i] = sign * parr[:, pos] * self.voxel_size[axis]
        return out.reshape(pixel_vector.shape)


# ---- standard images and ROI masks (utils/standard_images.py) -----------------------------------------
def full_like(image, fill_value, mode: str = "shape", dtype=None):
    from .image import ScalarImage

    if dtype is None:
        dtype = image.dtype
    if mode == "shape":
        return type(image)(np.full(image.shape, fill_value, dtype=dtype), **image.metadata())
    elif mode == "voxels":
        return ScalarImage(np.full(image.num_voxels, fill_value, dtype=dtype), **image.metadata())
    raise ValueError(f"mode {mode} not supported")


def zeros_like(image, mode: str = "shape", dtype=None):
    return full_like(image, 0, mode=mode, dtype=dtype)


def ones_like(image, mode: str = "shape", dtype=None):
    return full_like(image, 1, mode=mode, dtype=dtype)


def roi_to_mask(roi, reference_image, mode: str = "shape"):
    """Boolean full-domain mask of one bounding-box ROI, or the union of a list of them
    (utils/standard_images.py:73-137).  ``roi``: CoordinateArray (two opposite corners, metric) or
    VoxelArray (two opposite corners, voxels)."""
    if isinstance(roi, list):
        mask = zeros_like(reference_image, mode=mode, dtype=np.bool_)
        for item in roi:
            mask.img |= roi_to_mask(item, reference_image, mode=mode).img
        return mask
    mask = zeros_like(reference_image, mode=mode, dtype=np.bool_)
    if isinstance(roi, VoxelArray):
        box = roi
    elif isinstance(roi, (CoordinateArray, Coordinate)):
        box = reference_image.coordinatesystem.voxel(roi)
    else:
        raise TypeError(f"Unsupported ROI type: {type(roi)}")
    nv = reference_image.num_voxels
    row_min = int(max(0, np.min(box[:, 0])))
    row_max = min(int(np.max(box[:, 0])), nv[0])
    col_min = int(max(0, np.min(box[:, 1])))
    col_max = min(int(np.max(box[:, 1])), nv[1])
    if row_max > row_min and col_max > col_min:
        mask.img[row_min:row_max, col_min:col_max] = True
    return mask
