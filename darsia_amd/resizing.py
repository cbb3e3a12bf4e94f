"""``Resize`` / ``resize`` / ``equalize_voxel_size`` / ``uniform_refinement`` (reference: restoration/resize.py:18-376):
DarSIA's wrapper of ``cv2.resize`` -- "scale by downsampling" either side of the TV operator in the presets.

The reference splits the array into channels, calls ``cv2.resize(channel, dsize=, fx=, fy=[, interpolation=])`` and
merges; here all channels go through one device pass (``dxh_resize``).  OpenCV's arithmetic is third-party: the tap
and area tables are built below exactly as ``modules/imgproc/src/resize.cpp`` builds them (generic code paths) and the
kernel applies them in OpenCV's order of operations.  No OpenCV is available in this image, so parity with cv2 is
UNPINNED (float results may differ in the last bit from FMA-dispatching OpenCV builds); integer-factor
``inter_area``, ``inter_nearest`` and identity resizes are exact by construction.  float32 / float64 / uint8.

Interpolation: ``None`` -> OpenCV's default (INTER_LINEAR), ``"inter_area"``, ``"inter_linear"``, ``"inter_nearest"``.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np

from . import _lib
from .device import DeviceArray, host_empty

INTER_NEAREST, INTER_LINEAR, INTER_AREA = 0, 1, 3  # cv2's flag values
_DTYPES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.uint8): 2}
_F32 = np.float32


def _cv_round(x: float) -> int:
    return int(np.rint(x))


def _linear_taps(n_dst: int, n_src: int, scale: float, inv_scale: float, area_mode: bool):
    d = np.arange(n_dst, dtype=np.float64)
    if not area_mode:
        f = ((d + 0.5) * scale - 0.5).astype(_F32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(_F32)).astype(_F32)
    else:
        s = np.floor(d * scale).astype(np.int64)
        f = ((d + 1) - (s + 1) * inv_scale).astype(_F32)
        f = np.where(f <= 0, _F32(0), f - np.floor(f).astype(_F32)).astype(_F32)
    lo = s < 0
    s[lo], f[lo] = 0, 0
    hi = s >= n_src - 1
    s[hi], f[hi] = n_src - 1, 0
    return s.astype(np.int32), f


def _area_table(n_src: int, n_dst: int, scale: float):
    """computeResizeAreaTab in CSR form."""
    ptr, src, alpha = [0], [], []
    for d in range(n_dst):
        fs1 = d * scale
        fs2 = fs1 + scale
        cell = min(scale, n_src - fs1)
        s1, s2 = math.ceil(fs1), math.floor(fs2)
        s2 = min(s2, n_src - 1)
        s1 = min(s1, s2)
        if s1 - fs1 > 1e-3:
            src.append(s1 - 1)
            alpha.append((s1 - fs1) / cell)
        for s in range(s1, s2):
            src.append(s)
            alpha.append(1.0 / cell)
        if fs2 - s2 > 1e-3:
            src.append(s2)
            alpha.append(min(min(fs2 - s2, 1.0), cell) / cell)
        ptr.append(len(src))
    return np.asarray(ptr, np.int32), np.asarray(src, np.int32), np.asarray(alpha, np.float64).astype(_F32)


def cv_resize(img: np.ndarray, dsize=None, fx=None, fy=None, interpolation: Optional[int] = None) -> np.ndarray:
    """``cv2.resize`` of every channel of an (H, W) or (H, W, C) array (``dsize`` = (width, height) as in cv2).
    A ``DeviceArray`` in gives a ``DeviceArray`` out (only the small tap tables cross PCIe)."""
    if not isinstance(img, (np.ndarray, DeviceArray)) or img.ndim not in (2, 3):
        raise ValueError("img must be an (H, W) or (H, W, C) array")
    if img.dtype not in _DTYPES:
        raise TypeError(f"dtype {img.dtype} not supported on the device (float32, float64, uint8)")
    interpolation = INTER_LINEAR if interpolation is None else interpolation
    if interpolation not in (INTER_NEAREST, INTER_LINEAR, INTER_AREA):
        raise NotImplementedError(f"interpolation flag {interpolation}")
    h, w = img.shape[:2]
    if dsize is not None:
        dw, dh = int(dsize[0]), int(dsize[1])
        inv_x, inv_y = dw / w, dh / h
    else:
        if fx is None or fy is None or fx <= 0 or fy <= 0:
            raise ValueError("either dsize or positive fx, fy are needed")
        dw, dh, inv_x, inv_y = _cv_round(w * fx), _cv_round(h * fy), float(fx), float(fy)
    if dw < 1 or dh < 1:
        raise ValueError("empty destination size")
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    plan = _lib.ResizePlan()
    plan.struct_size = C.sizeof(_lib.ResizePlan)
    keep = []

    def put(name, arr):
        keep.append(arr)
        setattr(plan, name, arr.ctypes.data)

    if interpolation == INTER_NEAREST:
        plan.mode = 0
        put("xi", np.minimum(np.floor(np.arange(dw) * (1.0 / inv_x)).astype(np.int64), w - 1).astype(np.int32))
        put("yi", np.minimum(np.floor(np.arange(dh) * (1.0 / inv_y)).astype(np.int64), h - 1).astype(np.int32))
    elif interpolation == INTER_AREA and scale_x >= 1 and scale_y >= 1:
        isx, isy = _cv_round(scale_x), _cv_round(scale_y)
        eps = np.finfo(np.float64).eps
        if abs(scale_x - isx) < eps and abs(scale_y - isy) < eps:
            plan.mode, plan.isx, plan.isy = 2, isx, isy
        else:
            plan.mode = 3
            for ax, (n_src, n_dst, sc) in (("x", (w, dw, scale_x)), ("y", (h, dh, scale_y))):
                ptr, src, alpha = _area_table(n_src, n_dst, sc)
                put(ax + "ptr", ptr)
                put(ax + "src", src)
                put(ax + "alpha", alpha)
    else:
        plan.mode = 1
        area_mode = interpolation == INTER_AREA
        xi, xw = _linear_taps(dw, w, scale_x, inv_x, area_mode)
        yi, yw = _linear_taps(dh, h, scale_y, inv_y, area_mode)
        put("xi", xi), put("xw", xw), put("yi", yi), put("yw", yw)
    _lib.require_gpu()
    on_device = isinstance(img, DeviceArray)
    src = img if on_device else np.ascontiguousarray(img)
    chans = 1 if src.ndim == 2 else src.shape[2]
    out_shape = (dh, dw) if src.ndim == 2 else (dh, dw, chans)
    out = DeviceArray(out_shape, src.dtype, src.device) if on_device else host_empty(out_shape, src.dtype)
    _lib.check(_lib.load().dxh_resize(src.device if on_device else _lib.default_device(), _DTYPES[src.dtype], h, w, chans,
                                       _lib.ptr(src), dh, dw, C.byref(plan), _lib.ptr(out)))
    return out


class Resize:
    """Object for resizing 2-D arrays and Images (resize.py:18-251); same constructor keywords, incl. the
    ``key``-prefixed option names ``"<key>resize shape" / "resize" / "resize x" / "resize y" / "resize dtype" /
    "resize interpolation" / "resize conservative"``."""

    def __init__(self, ref_image=None, shape=None, fx=None, fy=None, interpolation: Optional[str] = None, dtype=None,
                 key: str = "", **kwargs) -> None:
        self.shape = kwargs.get(key + "resize shape", None) if shape is None else shape
        general_f = kwargs.get(key + "resize", None)
        self.fx = kwargs.get(key + "resize x", general_f) if fx is None else fx
        self.fy = kwargs.get(key + "resize y", general_f) if fy is None else fy
        self.dtype = kwargs.get(key + "resize dtype", None) if dtype is None else dtype
        if ref_image is not None:
            assert self.shape is None, "Provide only reference image or shape (not both)."
            self.shape = ref_image.num_voxels
        if self.shape is None:
            self.fx = 1 if self.fx is None else self.fx
            self.fy = 1 if self.fy is None else self.fy
            self.dsize = None
        else:
            self.dsize = tuple(reversed(self.shape))  # cv2 expects (width, height)
        pre = kwargs.get(key + "resize interpolation", None) if interpolation is None else interpolation
        if pre is None:
            self.interpolation = None
        elif pre == "inter_area":
            self.interpolation = INTER_AREA
        elif pre == "inter_linear":
            self.interpolation = INTER_LINEAR
        elif pre == "inter_nearest":
            self.interpolation = INTER_NEAREST
        else:
            raise NotImplementedError(f"Interpolation option {pre} is not implemented. Pick from:\n"
                                      " - inter_area\n - inter_linear\n - inter_nearest\n")
        conservative_key = key + "resize conservative"
        self._conservative_explicit = conservative_key in kwargs
        self.is_conservative = kwargs.get(conservative_key, False)

    def __str__(self) -> str:
        return "resize"

    def __call__(self, img, overwrite: bool = False):
        input_is_image = hasattr(img, "img") and hasattr(img, "metadata")
        if input_is_image:
            assert img.space_dim == 2
        extensive = input_is_image and bool(getattr(img, "extensive", False))
        # the reference copies the array first (resize.py:141); the device path never writes into its input, and a
        # 384 MB host copy costs more than the whole resize
        arr = img.img if input_is_image else img
        if isinstance(arr, DeviceArray):
            # device in, device out: the chain Resize -> TVD -> Resize of the presets
            # (presets/fluidflower/benchmarkco2model.py:78-84) stays in HBM
            if self.dtype is not None and np.dtype(self.dtype) != arr.dtype:
                raise TypeError("a device-resident array cannot change dtype in Resize")
            if arr.ndim not in (2, 3):
                raise ValueError("device-resident arrays with 2 or 3 axes are supported")
            resized = cv_resize(arr, dsize=self.dsize, fx=self.fx, fy=self.fy, interpolation=self.interpolation)
            if self.is_conservative:
                raise NotImplementedError("conservative rescaling of a device-resident array")
            return resized
        if self.dtype is not None:
            arr = arr.astype(self.dtype)
        original_shape = arr.shape
        multi = np.reshape(arr, (*original_shape[:2], -1))  # every axis past the second is a channel
        interpolation = INTER_AREA if extensive and self.interpolation is None else self.interpolation
        resized = cv_resize(np.ascontiguousarray(multi), dsize=self.dsize, fx=self.fx, fy=self.fy,
                            interpolation=interpolation)
        resized = np.reshape(resized, (*resized.shape[:2], *original_shape[2:]))
        if self.is_conservative or (extensive and not self._conservative_explicit):
            # in place, as the reference (resize.py:189-192): the result keeps the array's float type (and an integer
            # array refuses the float factor there as here)
            resized *= np.prod(arr.shape[:2]) / np.prod(resized.shape[:2])
        if input_is_image:
            return type(img)(resized, **img.metadata())
        return resized

    def save(self, path) -> None:
        np.savez(path, class_name=type(self).__name__, shape=self.shape, dsize=self.dsize, fx=self.fx, fy=self.fy,
                 interpolation=self.interpolation, dtype=self.dtype, is_conservative=self.is_conservative)

    def load(self, path) -> None:
        data = np.load(path, allow_pickle=True)

        def item(k):
            v = data[k]
            return v.item() if v.shape == () else tuple(v.tolist())

        self.shape = item("shape") if "shape" in data else None
        self.dsize = item("dsize") if "dsize" in data else None
        if self.dsize is None and self.shape is not None:
            self.dsize = tuple(reversed(self.shape))
        if self.shape is None and self.dsize is not None:
            self.shape = tuple(reversed(self.dsize))
        self.fx = item("fx") if "fx" in data else None
        self.fy = item("fy") if "fy" in data else None
        self.interpolation = item("interpolation") if "interpolation" in data else None
        self.dtype = item("dtype") if "dtype" in data else None
        self.is_conservative = item("is_conservative") if "is_conservative" in data else None
        self._conservative_explicit = "is_conservative" in data


def resize(image, ref_image=None, shape=None, fx=None, fy=None, interpolation: Optional[str] = None, dtype=None):
    """Function wrapper of ``Resize`` (resize.py:254-289)."""
    return Resize(ref_image=ref_image, shape=shape, fx=fx, fy=fy, interpolation=interpolation, dtype=dtype)(image)


def equalize_voxel_size(image, voxel_size: Optional[float] = None, **kwargs):
    """Keep the physical dimensions, unify the voxel side length (resize.py:292-319)."""
    if voxel_size is None:
        voxel_size = min(image.voxel_size)
    shape = tuple(int(d / voxel_size) for d in image.dimensions)
    return Resize(shape=shape, interpolation=kwargs.get("interpolation"))(image)


def uniform_refinement(image, levels: int):
    """Repeat (levels > 0) or pairwise-average (levels < 0) along the spatial axes (resize.py:322-376).  Plain index
    arithmetic on the host, as in the reference."""
    array = image.img.copy()
    for _ in range(abs(levels)):
        if levels > 0:
            for i in range(image.space_dim):
                array = np.repeat(array, 2, axis=i)
        elif levels < 0:
            for i in range(image.space_dim):
                def i_slice(item):
                    return tuple(item if i == j else slice(0, None) for j in range(image.space_dim))

                axis_length = image.img.shape[i]
                half = int(np.floor(axis_length) / 2)
                sub0, sub1 = array[i_slice(slice(0, None, 2))], array[i_slice(slice(1, None, 2))]
                array = np.multiply(0.5, sub0)
                array[i_slice(slice(0, half))] += np.multiply(0.5, sub1)
    return type(image)(array, **image.metadata())
