"""Dense colour-balance application on the GPU (reference:
corrections/color/colorbalance.py:10-241 and the tail of
``ColorCorrection.correct_array``, colorcorrection.py:388-394).

Only the per-pixel ``img @ A + b`` (+ clip, float32 cast) is device work; finding
the balance from a handful of swatches is a tiny host-side optimisation and is
kept as in the reference (SciPy)."""
from __future__ import annotations

import numpy as np

from . import _lib
from .device import DeviceArray, host_empty

_IN_CODES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.uint8): 2, np.dtype(np.uint16): 3}


def apply_affine_balance(img: np.ndarray, scaling: np.ndarray, translation=None, clip: bool = False) -> np.ndarray:
    """``img_as_float(img) @ scaling + translation`` (optionally clipped to [0, 1]) as float32.

    ``img``: (..., 3) array of uint8 / uint16 / float32 / float64 (integer types are scaled to [0, 1] like
    ``skimage.img_as_float``).  A ``DeviceArray`` in gives a ``DeviceArray`` out (nothing crosses PCIe)."""
    if not isinstance(img, (np.ndarray, DeviceArray)) or img.shape[-1] != 3:
        raise ValueError("img must be an array with a trailing axis of 3 colour channels")
    if img.dtype not in _IN_CODES:
        raise ValueError("Provide image in np.uint8, np.uint16, np.float32, or np.float64 format.")
    a = np.ascontiguousarray(scaling, dtype=np.float64)
    if a.shape != (3, 3):
        raise ValueError("scaling must be a 3x3 matrix")
    b = np.zeros(3) if translation is None else np.ascontiguousarray(translation, dtype=np.float64)
    if b.shape != (3,):
        raise ValueError("translation must have 3 entries")
    _lib.require_gpu()
    if isinstance(img, DeviceArray):
        src, device = img, img.device
        out = DeviceArray(src.shape, np.float32, device)
    else:
        src, device = np.ascontiguousarray(img), _lib.default_device()
        out = host_empty(src.shape, np.float32)
    _lib.check(_lib.load().dxh_affine_color(device, _IN_CODES[src.dtype], src.size // 3, _lib.ptr(src),
                                             _lib.ptr(a), _lib.ptr(b), int(bool(clip)), _lib.ptr(out)))
    return out


class BaseBalance:
    """Linear balance ``img @ balance_scaling`` (colorbalance.py:10-58); ``find_balance`` is host work."""

    def __init__(self) -> None:
        self.balance_scaling = np.eye(3)

    def apply_balance(self, img: np.ndarray) -> np.ndarray:
        return apply_affine_balance(img, self.balance_scaling)


class AffineBalance(BaseBalance):
    """General affine balance (colorbalance.py:140-196)."""

    def __init__(self) -> None:
        self.balance_scaling = np.eye(3)
        self.balance_translation = np.zeros(3)

    def find_balance(self, swatches_src: np.ndarray, swatches_dst: np.ndarray) -> None:
        """Least-squares affine fit by Powell's method, as in the reference (colorbalance.py:151-184)."""
        import scipy.optimize

        def objective(flat):
            return np.sum((swatches_src @ flat[:9].reshape((3, 3)) + flat[9:12] - swatches_dst) ** 2)

        res = scipy.optimize.minimize(objective, np.concatenate((self.balance_scaling.flatten(), self.balance_translation)),
                                      method="Powell", tol=1e-6, options={"maxiter": 1000, "disp": False})
        self.balance_scaling = res.x[:9].reshape((3, 3))
        self.balance_translation = res.x[9:12]

    def apply_balance(self, img: np.ndarray) -> np.ndarray:
        return apply_affine_balance(img, self.balance_scaling, self.balance_translation)
