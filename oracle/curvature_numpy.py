"""CPU restatement of the per-image part of DarSIA's ``CurvatureCorrection`` (corrections/shape/curvature.py) and
of the SciPy gather it rests on.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE (same rule as oracle/tvd_numpy.py).  PINNED: grid and corrected image of eight
configurations equal, bit for bit, what the reference's own CurvatureCorrection returns (tests/golden/curvature_golden.npz, made
by tests/golden/make_golden_curvature.py, which executes curvature.py / basecorrection.py; tests/test_curvature_golden_cpu.py).
The "crop" step: how the reference assembles it (configuration normal form, corner conventions, target size, dtype handling) is
pinned by tests/golden/make_golden_warp_assembly.py, which executes the reference's class and function with their two OpenCV
calls answered by oracle/warp_numpy.py; the arithmetic of cv2.warpPerspective itself stays unpinned -- the restatement describes the
classic fixed-point path of OpenCV 2.4 .. 4.10 (version label and upstream file / function names in oracle/warp_numpy.py's header).

* ``map_coordinates``: ``scipy.ndimage.map_coordinates(input2d, grid, order=0|1)`` with the default mode
  "constant", cval 0 (call site curvature.py:943-945).  SciPy is a third-party, unpinned dependency of the
  reference (pyproject.toml); it IS installed in this image (1.15.3), so the restatement below is checked against
  SciPy itself bit for bit in tests/test_oracle_curvature.py (float32/64 and uint8/16 images, float32/64 grids,
  points on and beyond every border).  Algorithm (ndimage/src/ni_interpolation.c, NI_GeometricTransform): a point
  with a coordinate < 0 or > n - 1 yields cval; order 1: start = floor(c), weights (1 - t, t) in double, taps past
  the end mirrored (their weight is 0), the four taps accumulated row-major as (v * w_row) * w_col; order 0: the tap
  at floor(c + 0.5); integer outputs are rounded half away from zero (unsigned: t > 0 ? t + 0.5 : 0).
* ``transform_coordinates`` / ``simple_curvature_correction`` / ``precompute_grid`` / ``transform_image``:
  curvature.py:734-957 on plain arrays (the polynomial bulge / stretch maps are NumPy float32 expressions and are
  repeated here operation by operation).  Parity: the reference module needs cv2 at import and its tests need image
  files that are not in the repository -> unpinned by the reference; pinned on SciPy for the gather.
"""
from __future__ import annotations

import numpy as np


def _mirror(idx, n):
    if n <= 1:
        return np.zeros_like(idx)
    s2 = 2 * n - 2
    idx = np.abs(idx) % s2
    return np.where(idx >= n, s2 - idx, idx)


def map_coordinates(a: np.ndarray, coords: np.ndarray, order: int = 1, cval: float = 0.0) -> np.ndarray:
    a = np.asarray(a)
    assert a.ndim == 2 and order in (0, 1)
    h, w = a.shape
    c0 = coords[0].astype(np.float64)
    c1 = coords[1].astype(np.float64)
    inside = (c0 >= 0) & (c0 <= h - 1) & (c1 >= 0) & (c1 <= w - 1)
    c0 = np.where(inside, c0, 0.0)
    c1 = np.where(inside, c1, 0.0)
    af = a.astype(np.float64)
    if order == 1:
        s0, s1 = np.floor(c0), np.floor(c1)
        t0, t1 = c0 - s0, c1 - s1
        w0, w1 = (1 - t0, t0), (1 - t1, t1)
        s0, s1 = s0.astype(np.int64), s1.astype(np.int64)
        t = np.zeros_like(c0)
        for i in range(2):
            for j in range(2):
                v = af[_mirror(s0 + i, h), _mirror(s1 + j, w)]
                v = v * w0[i]
                v = v * w1[j]
                t = t + v
    else:
        t = af[_mirror(np.floor(c0 + 0.5).astype(np.int64), h), _mirror(np.floor(c1 + 0.5).astype(np.int64), w)]
    t = np.where(inside, t, float(cval))
    if a.dtype.kind == "u":
        return np.where(t > 0, t + 0.5, 0).astype(a.dtype)
    if a.dtype.kind == "i":
        return np.where(t > 0, t + 0.5, t - 0.5).astype(a.dtype)
    return t.astype(a.dtype)


def transform_coordinates(X: np.ndarray, Y: np.ndarray, **kw):
    """curvature.py:857-919 (modifies and returns float32 meshes like the reference)."""
    hb, hs = kw.get("horizontal_bulge", 0.0), kw.get("horizontal_stretch", 0.0)
    hco = kw.get("horizontal_center_offset", 0)
    vb, vs = kw.get("vertical_bulge", 0.0), kw.get("vertical_stretch", 0.0)
    vco = kw.get("vertical_center_offset", 0)
    ny, nx = X.shape[:2]
    center = [round(nx / 2) + hco, round(ny / 2) + vco]
    X -= center[0]
    Y -= center[1]
    Xmod = (X + hb * np.multiply(X, (np.max(Y) - Y) * (Y - np.min(Y))) + hs * X * (np.max(X) - X) * (X - np.min(X)))
    Ymod = (Y + vb * np.multiply(Y, (np.max(X) - X) * (X - np.min(X))) + vs * Y * (np.max(Y) - Y) * (Y - np.min(Y)))
    Xmod += center[0]
    Ymod += center[1]
    return Xmod, Ymod


def transform_image(img: np.ndarray, grid: np.ndarray, shape, order: int = 1) -> np.ndarray:
    """curvature.py:921-957."""
    img3 = np.atleast_3d(img)
    out = np.zeros((*shape, img3.shape[2]), dtype=img3.dtype)
    for i in range(img3.shape[2]):
        out[:, :, i] = map_coordinates(img3[:, :, i], grid, order).reshape(shape)
    return np.squeeze(out)


def simple_curvature_correction(img: np.ndarray, order: int = 1, **kw) -> np.ndarray:
    """curvature.py:734-771."""
    ny, nx = img.shape[:2]
    X, Y = np.meshgrid(np.arange(nx, dtype=np.float32), np.arange(ny, dtype=np.float32))
    X, Y = transform_coordinates(X, Y, **kw)
    grid = np.array([Y.ravel(), X.ravel()])
    return transform_image(img, grid, X.shape[:2], order)


def precompute_grid(shape, config: dict, order: int = 1, crop=None):
    """curvature.py:775-825: the coordinate images go through init -> crop -> bulge -> stretch themselves.
    ``crop(pixels, **config["crop"])`` stands for darsia.extract_quadrilateral_ROI (cv2; not restated here)."""
    ny, nx = shape[:2]
    X, Y = np.meshgrid(np.arange(nx, dtype=np.float32), np.arange(ny, dtype=np.float32))
    res = {}
    for key, pixels in (("X", X), ("Y", Y)):
        if "init" in config:
            pixels = simple_curvature_correction(pixels, order, **config["init"])
        if "crop" in config:
            pixels = crop(pixels, **config["crop"])
        if "bulge" in config:
            pixels = simple_curvature_correction(pixels, order, **config["bulge"])
        if "stretch" in config:
            pixels = simple_curvature_correction(pixels, order, **config["stretch"])
        res[key] = pixels
    grid = np.array([res["Y"].ravel(), res["X"].ravel()])
    return grid, res["X"].shape[:2]
