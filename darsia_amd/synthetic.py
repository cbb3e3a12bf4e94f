"""Synthetic benchmark images: piecewise-constant blocks,
per-channel gains, Gaussian noise, clipped to [0, 1], float32."""
from __future__ import annotations

import numpy as np


def synthetic_image(h: int, w: int, c: int = 3, seed: int = 1234, dtype=np.float32, row0: int = 0,
                    noise: bool = True) -> np.ndarray:
    """Rows [row0, row0 + h) of the synthetic image of width ``w`` (row-slab friendly: the
    noise stream is seeded per slab start so every rank can generate its own rows)."""
    y = (np.arange(row0, row0 + h, dtype=np.float64))[:, None]
    x = (np.arange(w, dtype=np.float64))[None, :]
    base = 0.5 + 0.25 * np.sign(np.sin(2 * np.pi * x / 512.0)) * np.sign(np.sin(2 * np.pi * y / 384.0))
    gains = np.array([1.0, 0.8, 0.6, 0.9])[:c]
    img = base[..., None] * gains
    if noise:
        rng = np.random.default_rng([seed, row0])
        img = img + rng.normal(0.0, 0.05, size=img.shape)
    img = np.clip(img, 0.0, 1.0).astype(dtype)
    return img if c > 1 else img[..., 0]
