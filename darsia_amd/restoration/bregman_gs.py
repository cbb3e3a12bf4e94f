"""Split-Bregman TV denoising with scikit-image's Gauss-Seidel update on the GPU: what
``darsia.TVD(method="anisotropic bregman" | "isotropic bregman")`` computes through
``skimage.restoration.denoise_tv_bregman`` (reference call sites restoration/tvd.py:85-101,
utils/segmentation.py:117-120).

Signature of scikit-image >= 0.19 (the versions DarSIA's keyword ``max_num_iter`` needs).  Without
``channel_axis`` -- as DarSIA calls it -- an (H, W, C) array is ONE problem: the channels are independent 2-D
stencils that share the stop test.  ``weight``: the smaller, the smoother (``lam = 2 weight``).
The raster sweep's dependence is respected by sweeping anti-diagonals (csrc/dxh_bregman_gs.hip); results are
bit-identical to scikit-image 0.18.3's for float32 and float64.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .. import _lib
from ..device import DeviceArray, host_empty


def _img_as_float(image: np.ndarray) -> np.ndarray:
    if image.dtype.kind == "f":
        return image if image.dtype in (np.float32, np.float64) else image.astype(np.float64)
    if image.dtype == np.uint8:  # skimage multiplies by the reciprocal (util/dtype.py: np.multiply(image, 1. / imax))
        return np.multiply(image, 1.0 / 255, dtype=np.float64)
    if image.dtype == np.uint16:
        return np.multiply(image, 1.0 / 65535, dtype=np.float64)
    raise TypeError(f"dtype {image.dtype} not supported")


def denoise_tv_bregman(image: np.ndarray, weight: float = 5.0, max_num_iter: int = 100, eps: float = 1e-3,
                       isotropic: bool = True, *, channel_axis: Optional[int] = None, return_iters: bool = False):
    if isinstance(image, DeviceArray):
        # device in, device out (float images, no per-channel splitting: both are host work)
        if image.dtype not in (np.float32, np.float64) or image.ndim not in (2, 3) or channel_axis is not None:
            raise TypeError("a device-resident image must be float32 / float64 with 2 or 3 axes (channel_axis=None)")
        _lib.require_gpu()
        out = DeviceArray(image.shape, image.dtype, image.device)
        done, rmse = C.c_int(0), C.c_double(0.0)
        _lib.check(_lib.load().dxh_tv_bregman_gs(image.device, _lib.dtype_code(image.dtype), image.shape[0], image.shape[1],
                                                  image.shape[2] if image.ndim == 3 else 1, _lib.ptr(image), _lib.ptr(out),
                                                  float(weight), int(max_num_iter), float(eps), int(bool(isotropic)),
                                                  C.byref(done), C.byref(rmse)))
        if 1 in out.shape:
            out = out.reshape(tuple(n for n in out.shape if n != 1) or (1,))   # as scikit-image: singleton axes go
        return (out, done.value) if return_iters else out
    if not isinstance(image, np.ndarray):
        raise TypeError("image must be a numpy array")
    image = _img_as_float(image)
    if image.ndim not in (2, 3):
        raise ValueError("arrays with 2 or 3 axes are supported")
    if channel_axis is not None:
        # per-channel problems with their own stop tests (scikit-image loops over the channels)
        moved = np.moveaxis(image, channel_axis, -1)
        outs = [denoise_tv_bregman(np.ascontiguousarray(moved[..., c]), weight, max_num_iter, eps, isotropic,
                                   return_iters=True) for c in range(moved.shape[-1])]
        res = np.moveaxis(np.stack([o[0] for o in outs], axis=-1), -1, channel_axis)
        return (res, [o[1] for o in outs]) if return_iters else res
    _lib.require_gpu()
    src = np.ascontiguousarray(image)
    out = host_empty(src.shape, src.dtype)
    dims = src.shape[2] if src.ndim == 3 else 1
    done = C.c_int(0)
    rmse = C.c_double(0.0)
    _lib.check(_lib.load().dxh_tv_bregman_gs(_lib.default_device(), _lib.dtype_code(src.dtype), src.shape[0], src.shape[1],
                                              dims, _lib.ptr(src), _lib.ptr(out), float(weight), int(max_num_iter),
                                              float(eps), int(bool(isotropic)), C.byref(done), C.byref(rmse)))
    out = np.squeeze(out)  # as scikit-image: every singleton axis goes
    return (out, done.value) if return_iters else out
