"""CPU restatement (NumPy) of the dense part of DarSIA's image registration:
the piecewise perspective warp and the thin-plate-spline displacement field.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

PARITY UNPINNED.  The reference does this work in OpenCV (``cv2.getPerspectiveTransform``,
``cv2.warpPerspective(..., INTER_LINEAR)``; corrections/shape/piecewiseperspective.py:71-136) and in
SciPy (``RBFInterpolator``; multi_image_analysis/translationanalysis.py:285-292).  opencv-python is an
UNPINNED dependency (pyproject.toml:41) and is NOT installed in this environment, and the reference's
only numeric test of this path needs image files that are not in the repository
(tests/unit/test_image_registration.py:118-156 -> xfail).  What is restated here is OpenCV's published
algorithm for the float32 bilinear perspective warp (modules/imgproc/src/imgwarp.cpp,
WarpPerspectiveInvoker + remapBilinear):

  * M = getPerspectiveTransform(src, dst): the 8x8 linear system for the homography, solved in double;
    warpPerspective inverts it (double) because WARP_INVERSE_MAP is not set;
  * the destination is processed in blocks of bw0 x bh0 pixels (bh0 = min(16, h), bw0 = min(1024/bh0, w),
    bh0 = min(1024/bw0, h)); for the block column bx and row y:  X0 = M0*bx + M1*y + M2 (same for Y0, W0),
    and per pixel x1 = x - bx:  W = 32/(W0 + M6*x1) (0 if the denominator is 0),
    X = round_half_even((X0 + M0*x1)*W) clamped to int32 -- source coordinates in 1/32 pixel units;
  * integer part (X >> 5, saturated to int16) and fraction (X & 31) index a float table of bilinear
    weights  w[ky][kx] = {1 - fy, fy}[ky] * {1 - fx, fx}[kx]  (fx = frac/32 as float32);
  * value = S00*w00 + S01*w01 + S10*w10 + S11*w11 in float32, left to right, taps outside the source
    replaced by the constant border 0.

VERSION LABEL (what this restatement describes; from the builder's reading of the upstream sources -- not checkable here,
there is neither cv2 nor network): the classic path of OpenCV 2.4.x .. 4.10.x (opencv-python <= 4.10.0.84) --
``modules/imgproc/src/imgwarp.cpp``: ``cv::getPerspectiveTransform`` (8 x 8 system, ``solve(..., DECOMP_SVD)`` in double),
``cv::warpPerspective`` -> ``WarpPerspectiveInvoker::operator()`` (block sizes ``BLOCK_SZ = 32`` -> bh0 / bw0 as above, the
``INTER_BITS = 5`` fixed-point coordinates with ``saturate_cast<int>`` of the rounded product, ``INTER_TAB_SIZE = 32``) ->
``cv::remap`` -> ``remapBilinear<Cast<float, float>, RemapNoVec, float>`` with the float table ``BilinearTab_f``
(``initInterTab2D``) and ``BORDER_CONSTANT``.  The SSE4.1 / AVX2 / NEON line kernels of those releases compute the same
fixed-point coordinates.  OpenCV 4.11 (2025) and later route INTER_LINEAR warps of CV_8U / CV_16U / CV_32F images with 1, 3 or 4
channels through new kernels (``modules/imgproc/src/warp_kernels.simd.hpp``, ``warpPerspectiveLinearInvoker_*``) that
interpolate at the exact floating-point source position -- no 1/32-pixel quantisation --, so an installation with such a release
differs from this restatement (and from the reference's own results under an older OpenCV) by up to ~1/64 pixel of sampling
position.  The reference pins no OpenCV version (pyproject.toml:41), so WHICH of the two a DarSIA installation computes depends
on the day it was installed; this repository reproduces the classic one.

The thin-plate-spline evaluation is checked against SciPy itself (available here), see tests.
"""
from __future__ import annotations

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def get_perspective_transform(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """3x3 homography mapping the 4 points ``src`` to ``dst`` (both (4, 2), x/y order, float32 like cv2)."""
    src = np.asarray(src, dtype=np.float32).astype(np.float64)
    dst = np.asarray(dst, dtype=np.float32).astype(np.float64)
    a = np.zeros((8, 8))
    b = np.zeros(8)
    for i in range(4):
        x, y = src[i]
        X, Y = dst[i]
        a[i] = [x, y, 1, 0, 0, 0, -x * X, -y * X]
        a[i + 4] = [0, 0, 0, x, y, 1, -x * Y, -y * Y]
        b[i] = X
        b[i + 4] = Y
    c = np.linalg.solve(a, b)
    return np.append(c, 1.0).reshape(3, 3)


def _block_sizes(w: int, h: int):
    bh0 = min(16, h)
    bw0 = min(1024 // bh0, w)
    bh0 = min(1024 // bw0, h)
    return bw0, bh0


def source_coordinates(minv: np.ndarray, w: int, h: int):
    """Fixed-point source coordinates (X, Y in 1/32 pixel units, int64) of every destination pixel."""
    m = np.asarray(minv, dtype=np.float64).ravel()
    bw0, _ = _block_sizes(w, h)
    x = np.arange(w)
    bx = (x // bw0) * bw0
    x1 = (x - bx).astype(np.float64)
    bx = bx.astype(np.float64)
    y = np.arange(h, dtype=np.float64)[:, None]
    X0 = m[0] * bx[None, :] + m[1] * y + m[2]
    Y0 = m[3] * bx[None, :] + m[4] * y + m[5]
    W0 = m[6] * bx[None, :] + m[7] * y + m[8]
    W = W0 + m[6] * x1[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        W = np.where(W != 0, INTER_TAB_SIZE / W, 0.0)
    lo, hi = float(np.iinfo(np.int32).min), float(np.iinfo(np.int32).max)
    fX = np.maximum(lo, np.minimum(hi, (X0 + m[0] * x1[None, :]) * W))
    fY = np.maximum(lo, np.minimum(hi, (Y0 + m[3] * x1[None, :]) * W))
    return np.rint(fX).astype(np.int64), np.rint(fY).astype(np.int64)


def warp_perspective(src: np.ndarray, m: np.ndarray, w: int, h: int) -> np.ndarray:
    """cv2.warpPerspective(src, m, (w, h), flags=INTER_LINEAR) for float32 images, constant border 0."""
    return warp_with_inverse(src, np.linalg.inv(np.asarray(m, dtype=np.float64)), w, h)


def warp_with_inverse(src: np.ndarray, minv: np.ndarray, w: int, h: int) -> np.ndarray:
    """The same with the inverted matrix given (what warpPerspective works with after its own inversion): lets the tests
    feed degenerate maps -- W = 0, denormal or huge denominators -- that no invertible homography produces."""
    src = np.asarray(src, dtype=np.float32)
    sh, sw = src.shape[:2]
    X, Y = source_coordinates(minv, w, h)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    fx = ((X & (INTER_TAB_SIZE - 1)).astype(np.float32)) * np.float32(1.0 / INTER_TAB_SIZE)
    fy = ((Y & (INTER_TAB_SIZE - 1)).astype(np.float32)) * np.float32(1.0 / INTER_TAB_SIZE)
    wx = [np.float32(1.0) - fx, fx]
    wy = [np.float32(1.0) - fy, fy]
    extra = src.shape[2:]
    out = np.zeros((h, w) + extra, np.float32)
    first = True
    for ky in (0, 1):
        for kx in (0, 1):
            yy, xx = sy + ky, sx + kx
            ok = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
            tap = np.zeros((h, w) + extra, np.float32)
            tap[ok] = src[yy[ok], xx[ok]]
            wgt = (wy[ky] * wx[kx]).astype(np.float32)
            term = tap * (wgt[..., None] if extra else wgt)
            out = term if first else out + term
            first = False
    return out.astype(np.float32)


def find_and_warp(base_img: np.ndarray, patch_rois, local_corners, global_corners, displacement, reverse=False):
    """piecewiseperspective.py:35-145 on plain arrays.

    ``patch_rois[i][j]``: (row slice, col slice) of patch (i, j) in ``base_img``;
    ``local_corners[i][j]`` / ``global_corners[i][j]``: (4, 2) corner voxels (row, col) in patch / image
    coordinates; ``displacement(points_xy)`` -> (2, n) array like the reference's interpolator pair."""
    h, w = base_img.shape[:2]
    canvas = np.zeros(base_img.shape, np.float32)
    ni, nj = len(patch_rois), len(patch_rois[0])
    for i in range(ni):
        for j in range(nj):
            gc = np.fliplr(np.asarray(global_corners[i][j], dtype=np.float64))
            lc = np.fliplr(np.asarray(local_corners[i][j], dtype=np.float64))
            pts_dst = gc + (-1.0 if reverse else 1.0) * np.asarray(displacement(gc)).T
            origin = np.array([max(0, np.min(pts_dst[:, 0])), max(0, np.min(pts_dst[:, 1]))]).astype(np.int32)
            pts_dst_eff = pts_dst - origin
            w_eff = int(np.ceil(min(w, np.max(pts_dst[:, 0])) - max(0, origin[0])))
            h_eff = int(np.ceil(min(h, np.max(pts_dst[:, 1])) - max(0, origin[1])))
            if min(h_eff, w_eff) <= 1:
                continue
            r0, r1 = max(0, int(origin[1])), min(h, int(origin[1] + h_eff))
            c0, c1 = max(0, int(origin[0])), min(w, int(origin[0] + w_eff))
            m = get_perspective_transform(lc.astype(np.float32), pts_dst_eff.astype(np.float32))
            patch = base_img[patch_rois[i][j]].astype(np.float32)
            tmp = warp_perspective(patch, m, w_eff, h_eff)
            canvas[r0:r1, c0:c1] += tmp[: r1 - int(origin[1]), : c1 - int(origin[0])]
    return canvas.astype(base_img.dtype)


def thin_plate_spline_eval(points, centers, coeffs, shift, scale, powers):
    """SciPy RBFInterpolator evaluation for kernel='thin_plate_spline', epsilon=1:
    sum_i c_i r_i^2 log r_i + polynomial in (x - shift)/scale."""
    x = np.asarray(points, dtype=np.float64)
    y = np.asarray(centers, dtype=np.float64)
    r = np.sqrt(((x[:, None, :] - y[None, :, :]) ** 2).sum(-1))
    with np.errstate(divide="ignore", invalid="ignore"):
        phi = np.where(r == 0, 0.0, r**2 * np.log(r))
    xhat = (x - shift) / scale
    poly = np.prod(xhat[:, None, :] ** np.asarray(powers)[None, :, :], axis=-1)
    p = y.shape[0]
    return phi @ coeffs[:p] + poly @ coeffs[p:]
