"""dtype conversion (reference: utils/dtype.py:9-30, which defers to skimage.img_as_*).

Only the float <-> float conversions the TV path needs are kept; they are plain
casts, which is what ``skimage.img_as_float32/64`` do for float input."""
from __future__ import annotations

from warnings import warn

import numpy as np


def convert_dtype(img: np.ndarray, dtype) -> np.ndarray:
    dtype = np.dtype(dtype)
    if dtype in (np.dtype(np.float32), np.dtype(np.float64)):
        return img.astype(dtype, copy=False)
    if dtype in (np.dtype(np.uint8), np.dtype(np.uint16)):
        # skimage.img_as_ubyte / img_as_uint semantics for float input in [0, 1]: scale, round, clip
        info = np.iinfo(dtype)
        if img.dtype.kind == "f":
            if img.min() < -1.0 or img.max() > 1.0:
                raise ValueError("Images of type float must be between -1 and 1.")
            return np.clip(np.rint(img * info.max), 0, info.max).astype(dtype)
        return img.astype(dtype)
    warn(f"{dtype} is not a supported dtype. Returning {img.dtype} image.")
    return img
