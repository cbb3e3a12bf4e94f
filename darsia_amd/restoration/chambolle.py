"""Chambolle TV denoising on the GPU: what ``darsia.TVD()`` (default method
"chambolle") computes through ``skimage.restoration.denoise_tv_chambolle``
(reference call site restoration/tvd.py:77-83).

As in that call, no channel axis is passed: an (H, W, 3) image is treated as a
3-D volume.  Arrays with 2 or 3 axes, float32/float64 (other dtypes are
converted like ``img_as_float`` does for the common integer image types).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib
from ..device import DeviceArray, host_empty


def denoise_tv_chambolle(image: np.ndarray, weight: float = 0.1, eps: float = 2e-4, max_num_iter: int = 200,
                         return_iters: bool = False):
    if isinstance(image, DeviceArray):
        # device in, device out (float images only: the integer conversions above are host work)
        if image.dtype not in (np.float32, np.float64) or image.ndim not in (2, 3):
            raise TypeError("a device-resident image must be float32 / float64 with 2 or 3 axes")
        out = DeviceArray(image.shape, image.dtype, image.device)
        done = C.c_int(0)
        _lib.check(_lib.load().dxh_chambolle(image.device, _lib.dtype_code(image.dtype), image.ndim, image.shape[0],
                                              image.shape[1], image.shape[2] if image.ndim == 3 else 1,
                                              _lib.ptr(image), _lib.ptr(out), float(weight), float(eps),
                                              int(max_num_iter), C.byref(done)))
        return (out, done.value) if return_iters else out
    if not isinstance(image, np.ndarray):
        raise TypeError("image must be a numpy array")
    if image.dtype.kind != "f":
        if image.dtype == np.uint8:
            image = np.multiply(image, 1.0 / 255, dtype=np.float64)  # skimage's img_as_float multiplies
        elif image.dtype == np.uint16:
            image = np.multiply(image, 1.0 / 65535, dtype=np.float64)
        else:
            raise TypeError(f"dtype {image.dtype} not supported")
    if image.dtype not in (np.float32, np.float64):
        image = image.astype(np.float64)
    if image.ndim not in (2, 3):
        raise ValueError("arrays with 2 or 3 axes are supported")
    _lib.require_gpu()
    src = np.ascontiguousarray(image)
    out = host_empty(src.shape, src.dtype)
    n2 = src.shape[2] if src.ndim == 3 else 1
    done = C.c_int(0)
    _lib.check(_lib.load().dxh_chambolle(_lib.default_device(), _lib.dtype_code(src.dtype), src.ndim, src.shape[0],
                                          src.shape[1], n2, _lib.ptr(src), _lib.ptr(out), float(weight), float(eps),
                                          int(max_num_iter), C.byref(done)))
    if return_iters:
        return out, done.value
    return out
