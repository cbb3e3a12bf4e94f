"""Synthetic benchmark images: piecewise-constant blocks,
per-channel gains, Gaussian noise N(0, 0.05^2), clipped to [0, 1], float32.

The noise stream is seeded per block of 256 global rows, so any rank can
generate any row range of the same global image (row slabs with halos)."""
from __future__ import annotations

import numpy as np

_BLOCK = 256


def synthetic_image(h: int, w: int, c: int = 3, seed: int = 1234, dtype=np.float32, row0: int = 0) -> np.ndarray:
    """Rows [row0, row0 + h) of the global synthetic image of width ``w`` with ``c`` channels
    (``c == 1`` returns a 2-D array)."""
    y = np.arange(row0, row0 + h, dtype=np.float64)[:, None]
    x = np.arange(w, dtype=np.float64)[None, :]
    base = 0.5 + 0.25 * np.sign(np.sin(2 * np.pi * x / 512.0)) * np.sign(np.sin(2 * np.pi * y / 384.0))
    gains = np.array([1.0, 0.8, 0.6, 0.9])[:c]
    img = np.empty((h, w, c), dtype=dtype)
    r = row0
    while r < row0 + h:
        blk = r // _BLOCK
        r_end = min((blk + 1) * _BLOCK, row0 + h)
        rng = np.random.default_rng([seed, blk])
        noise = rng.normal(0.0, 0.05, size=(_BLOCK, w, c)).astype(np.float32)
        lo, hi = r - blk * _BLOCK, r_end - blk * _BLOCK
        rows = slice(r - row0, r_end - row0)
        img[rows] = np.clip(base[rows, :, None] * gains + noise[lo:hi], 0.0, 1.0)
        r = r_end
    return img if c > 1 else np.ascontiguousarray(img[..., 0])
