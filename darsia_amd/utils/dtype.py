"""dtype conversion (reference: utils/dtype.py:9-30, which defers to ``skimage.img_as_ubyte / img_as_uint /
img_as_float32 / img_as_float64``).

scikit-image is a third-party dependency of the reference (unpinned, absent on the GPU box), so its conversion
rules (skimage/util/dtype.py ``_convert``, 0.18.3) are restated here for the image dtypes of this path -- uint8,
uint16, float32, float64 -- and pinned by goldens generated through the reference's own ``convert_dtype`` with the
real scikit-image behind it (tests/golden/make_golden_models.py, tests/test_models_golden_cpu.py):

* same dtype: the input itself (no copy);
* float -> float: a cast;
* float -> unsigned: values must lie in [-1, 1] (``ValueError`` otherwise); multiply by the integer maximum in the
  INPUT's float type, round half to even, clip to [0, max], cast;
* unsigned -> float: multiply by ``1 / max_in`` in the OUTPUT float type;
* uint8 -> uint16: times 257 (exact); uint16 -> uint8: floor division by 256, unless every value already fits in
  8 bits, in which case scikit-image only casts (and warns).
"""
from __future__ import annotations

from warnings import warn

import numpy as np

_SUPPORTED = (np.dtype(np.uint8), np.dtype(np.uint16), np.dtype(np.float32), np.dtype(np.float64))


def convert_dtype(img: np.ndarray, dtype) -> np.ndarray:
    try:
        dtype = np.dtype(dtype)
    except TypeError:
        dtype = None
    if dtype not in _SUPPORTED:
        warn(f"{dtype} is not a supported dtype. Returning {img.dtype} image.")
        return img
    img = np.asarray(img)
    if img.dtype == dtype:
        return img
    if img.dtype not in _SUPPORTED:
        raise ValueError(f"Can not convert from {img.dtype} to {dtype}.")
    kin, kout = img.dtype.kind, dtype.kind
    if kin == "f":
        if kout == "f":
            return img.astype(dtype)
        if np.min(img) < -1.0 or np.max(img) > 1.0:
            raise ValueError("Images of type float must be between -1 and 1.")
        imax = np.iinfo(dtype).max
        out = np.multiply(img, imax, dtype=img.dtype)
        np.rint(out, out=out)
        np.clip(out, 0, imax, out=out)
        return out.astype(dtype)
    imax_in = np.iinfo(img.dtype).max
    if kout == "f":
        return np.asarray(np.multiply(img, 1.0 / imax_in, dtype=dtype), dtype)
    if img.dtype.itemsize < dtype.itemsize:          # uint8 -> uint16
        out = np.empty(img.shape, dtype)
        np.multiply(img, (2**16 - 1) // (2**8 - 1), out=out, dtype=dtype)
        return out
    if img.size and img.max() < 2**8:                # uint16 that already fits: cast only, as scikit-image does
        warn(f"Downcasting {img.dtype} to uint8 without scaling because max value {img.max()} fits in uint8")
        return img.astype(dtype)
    out = np.empty(img.shape, dtype)
    np.floor_divide(img, 2**8, out=out, dtype=img.dtype, casting="unsafe")
    return out
