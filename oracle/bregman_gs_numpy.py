"""CPU restatement of scikit-image's split-Bregman TV denoising with a Gauss-Seidel u-update -- the arithmetic
behind DarSIA's ``TVD(method="anisotropic bregman" | "isotropic bregman")`` (restoration/tvd.py:85-101) and the
``"tvd"`` smoothing of utils/segmentation.py:117-120.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE (same rule as oracle/tvd_numpy.py).

scikit-image is a third-party, un-vendored, UNPINNED dependency of the reference (pyproject.toml:44) and this
kernel ships compiled only (``skimage/restoration/_denoise_cy`` -- no source in the image).  The algorithm is the
published one (Goldstein & Osher 2009; Getreuer, IPOL 2012) as scikit-image implements it; the details that are
not in the papers were fixed by probing scikit-image 0.18.3 until every candidate detail matched BIT FOR BIT
(float64 and float32, both variants, with and without early stop):

* the image sits in a frame of one pixel that is filled ONCE, before the first sweep, as
  ``top = image[1], left = image[:, 1], bottom = image[-1], right = image[:, -1]`` (sic: reflected at the top and
  left, replicated at the bottom and right), corners 0; the frame of d and b stays 0;
* channels (third axis) are independent 2-D problems swept one after the other; ``lam = 2 weight``,
  ``norm = weight + 4 lam``; raster order, in place::

      u      = (lam (u[r+1,c] + u[r-1,c] + u[r,c+1] + u[r,c-1] + dx[r,c-1] - dx[r,c] + dy[r-1,c] - dy[r,c]
                     - bx[r,c-1] + bx[r,c] - by[r-1,c] + by[r,c]) + weight f[r,c]) / norm       (left to right)
      ux, uy = forward differences of the OLD u[r,c] against the not yet updated u[r,c+1], u[r+1,c]
      anisotropic: dx = shrink(ux + bx, 1/lam);   isotropic: s = |(ux+bx, uy+by)|, dx = s lam tx / (s lam + 1)
      bx    += ux - dx   (likewise y)

  In the float32 build the literals ``1`` of ``1/lam`` and ``s lam + 1`` are C doubles, so those sub-expressions
  (and ``sqrt``) are evaluated in double and rounded once;
* the sweep repeats while ``i < max_iter and sqrt(sum (u_new - u_old)^2 / size) > eps`` (all channels jointly).

Cells on one anti-diagonal r + c = const do not depend on each other, which is how the device sweeps
(`sweep="wavefront"` here evaluates them together and must -- and does -- give identical bits).

Parity status: pinned against scikit-image 0.18.3 outputs (tests/golden/skimage_0_18_3_golden.npz); the reference's
own tests hold no numeric result for these methods ("parity unpinned by the reference").  DarSIA calls
``denoise_tv_bregman(img, weight=, max_num_iter=, eps=, isotropic=)`` without a channel axis, i.e. ONE call over all
channels with the joint stop test (``multichannel=False`` in 0.18.3).
"""
from __future__ import annotations

import numpy as np


def img_as_float(image: np.ndarray) -> np.ndarray:
    """skimage.util.img_as_float for the dtypes DarSIA feeds: floats pass, uint8/uint16 are MULTIPLIED by
    1/255 resp. 1/65535 in float64 (skimage.util.dtype._convert uses ``np.multiply(image, 1. / imax_in)``)."""
    image = np.asarray(image)
    if image.dtype.kind == "f":
        return image
    if image.dtype == np.uint8:
        return np.multiply(image, 1.0 / 255, dtype=np.float64)
    if image.dtype == np.uint16:
        return np.multiply(image, 1.0 / 65535, dtype=np.float64)
    raise TypeError(f"unsupported dtype {image.dtype}")


def _shrink_terms(T, lam, ux, uy, bxx, byy, isotropic):
    """d-subproblem on arrays (or scalars) of dtype T, with the double-literal evaluation of the float32 build."""
    f64 = np.float64
    if isotropic:
        tx = ux + bxx
        ty = uy + byy
        s = np.sqrt((tx * tx + ty * ty).astype(f64)).astype(T)
        sl = s * lam
        den = sl.astype(f64) + 1.0
        dxx = ((sl * tx).astype(f64) / den).astype(T)
        dyy = ((sl * ty).astype(f64) / den).astype(T)
    else:
        th = 1.0 / f64(lam)

        def shr(s):
            sd = s.astype(f64)
            return np.where(sd > th, (sd - th).astype(T), np.where(sd < -th, (sd + th).astype(T), T(0))).astype(T)

        dxx = shr(ux + bxx)
        dyy = shr(uy + byy)
    return dxx, dyy


def denoise_tv_bregman(image, weight=5.0, max_num_iter=100, eps=1e-3, isotropic=True, sweep="wavefront",
                       return_iterations=False):
    image = np.atleast_3d(img_as_float(image))
    T = image.dtype.type
    rows, cols, dims = image.shape
    out = np.zeros((rows + 2, cols + 2, dims), image.dtype)
    dx, dy, bx, by = (np.zeros_like(out) for _ in range(4))
    weight = T(weight)
    lam = T(2) * weight
    norm = weight + T(4) * lam
    out[1:-1, 1:-1] = image
    if rows > 1:
        out[0, 1:-1] = image[1]
    if cols > 1:
        out[1:-1, 0] = image[:, 1]
    out[-1, 1:-1] = image[-1]
    out[1:-1, -1] = image[:, -1]
    total = rows * cols * dims
    i, rmse = 0, np.inf
    with np.errstate(all="ignore"):
        while i < max_num_iter and rmse > eps:
            sq = 0.0
            if sweep == "raster":
                for k in range(dims):
                    for r in range(1, rows + 1):
                        for c in range(1, cols + 1):
                            sq += _cell(T, out, dx, dy, bx, by, image, lam, weight, norm, np.array([r]), np.array([c]), k,
                                        isotropic)
            else:
                for d in range(2, rows + cols + 1):
                    r = np.arange(max(1, d - cols), min(rows, d - 1) + 1)
                    c = d - r
                    for k in range(dims):
                        sq += _cell(T, out, dx, dy, bx, by, image, lam, weight, norm, r, c, k, isotropic)
            rmse = np.sqrt(sq / total)
            i += 1
    res = np.squeeze(out[1:-1, 1:-1])
    return (res, i) if return_iterations else res


def _cell(T, out, dx, dy, bx, by, image, lam, weight, norm, r, c, k, isotropic):
    uprev = out[r, c, k]
    ux = out[r, c + 1, k] - uprev
    uy = out[r + 1, c, k] - uprev
    acc = out[r + 1, c, k] + out[r - 1, c, k]
    acc = acc + out[r, c + 1, k]
    acc = acc + out[r, c - 1, k]
    acc = acc + dx[r, c - 1, k]
    acc = acc - dx[r, c, k]
    acc = acc + dy[r - 1, c, k]
    acc = acc - dy[r, c, k]
    acc = acc - bx[r, c - 1, k]
    acc = acc + bx[r, c, k]
    acc = acc - by[r - 1, c, k]
    acc = acc + by[r, c, k]
    unew = (lam * acc + weight * image[r - 1, c - 1, k]) / norm
    out[r, c, k] = unew
    t = (unew - uprev).astype(np.float64)
    dxx, dyy = _shrink_terms(T, lam, ux, uy, bx[r, c, k], by[r, c, k], isotropic)
    dx[r, c, k] = dxx
    dy[r, c, k] = dyy
    bx[r, c, k] += ux - dxx
    by[r, c, k] += uy - dyy
    return float(np.sum(t * t))
