"""CPU restatement of ``cv2.resize`` as DarSIA's ``Resize`` uses it (restoration/resize.py:18-200): one call per
channel with ``dsize`` or ``fx, fy`` and interpolation INTER_NEAREST / INTER_LINEAR (OpenCV's default) / INTER_AREA.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE (same rule as oracle/tvd_numpy.py).

PARITY UNPINNED: OpenCV (``opencv-python``, unpinned in the reference's pyproject.toml:41) is not installed in this
image and the reference's tests hold no numeric result for ``Resize``.  What follows restates the published
algorithm of OpenCV 4.x ``modules/imgproc/src/resize.cpp`` (generic, non-SIMD code paths):

* destination size: ``dsize`` if given, else ``(cvRound(w fx), cvRound(h fy))`` (round half to even);
  ``inv_scale = dsize / ssize`` when ``dsize`` is given, else ``fx, fy``; ``scale = 1 / inv_scale`` (double).
* INTER_NEAREST: ``sx = min(floor(dx * (1 / inv_scale_x)), w - 1)``, likewise y.
* INTER_LINEAR: ``f = float((dx + 0.5) scale - 0.5)``, ``sx = floor(f)``, ``f -= sx`` (float32); ``sx < 0`` ->
  ``sx = 0, f = 0``; ``sx >= w - 1`` -> ``sx = w - 1, f = 0``; weights ``(1 - f, f)`` in float32.
  float32 images: rows first, ``S[sx] a0 + S[sx+1] a1`` in float32, then ``R0 b0 + R1 b1`` in float32;
  float64 images: the same with float32 weights and double arithmetic;
  uint8 images: weights ``short(round(w * 2048))``, rows ``S[sx] a0 + S[sx+1] a1`` in int32, columns
  ``(((b0 (R0 >> 4)) >> 16) + ((b1 (R1 >> 4)) >> 16) + 2) >> 2``.
* INTER_AREA, shrinking in both directions: with integer scales the mean of the ``sx x sy`` block --
  ``sum`` accumulated in float32 (float32 / uint8 images: uint8 sums are integers; float64: double) times
  ``float(1 / (sx sy))``; uint8 2 x 2 blocks use OpenCV's SIMD form ``(s + 2) >> 2``.  Otherwise the
  area-weight tables of ``computeResizeAreaTab`` (cells of width ``scale``, partial cells weighted by their
  overlap, entries smaller than 1e-3 dropped, weights float32), accumulated row by row:
  ``buf = sum_x S alpha_x`` then ``sum (+)= beta_y buf`` in float32 (double for float64).
  INTER_AREA when enlarging falls back to the bilinear scheme with ``sx = floor(dx scale)``,
  ``f = (dx + 1) - (sx + 1) inv_scale`` clipped to [0, 1).
* uint8 results: ``cvRound`` (half to even) and saturation.

VERSION LABEL (from the builder's reading of the upstream sources; not checkable here -- no cv2, no network): OpenCV
3.4.x .. 4.x (through 4.10; the file is unchanged in these respects in later 4.x as far as known) --
``modules/imgproc/src/resize.cpp``: ``cv::resize`` size / scale rules, ``resizeNN`` (``resizeNNInvoker``),
``resizeGeneric_<HResizeLinear, VResizeLinear>`` with ``INTER_RESIZE_COEF_BITS = 11`` for 8-bit images and float weights
otherwise, ``resizeAreaFast_`` (integer scales; the 8-bit 2 x 2 SIMD form ``(s + 2) >> 2``) and ``resizeArea_`` with
``computeResizeAreaTab``.  ``INTER_LINEAR_EXACT`` / ``INTER_NEAREST_EXACT`` (bit-exact integer paths added in 3.4 / 4.5) are
different flags and not what DarSIA passes (restoration/resize.py:80-96).
"""
from __future__ import annotations

import math

import numpy as np

INTER_NEAREST, INTER_LINEAR, INTER_AREA = 0, 1, 3
F32 = np.float32


def cv_round(x: float) -> int:
    return int(np.rint(x))


def dest_size(shape, dsize=None, fx=None, fy=None):
    h, w = shape[:2]
    if dsize is not None:
        dw, dh = int(dsize[0]), int(dsize[1])
        return dw, dh, dw / w, dh / h
    return cv_round(w * fx), cv_round(h * fy), float(fx), float(fy)


def _linear_taps(n_dst, n_src, scale, inv_scale, area_mode):
    idx = np.empty(n_dst, np.int64)
    w1 = np.empty(n_dst, F32)
    for d in range(n_dst):
        if not area_mode:
            f = F32((d + 0.5) * scale - 0.5)
            s = int(math.floor(f))
            f = F32(f - F32(s))
        else:
            s = int(math.floor(d * scale))
            f = F32((d + 1) - (s + 1) * inv_scale)
            f = F32(0) if f <= 0 else F32(f - F32(math.floor(f)))
        if s < 0:
            s, f = 0, F32(0)
        if s >= n_src - 1:
            s, f = n_src - 1, F32(0)
        idx[d], w1[d] = s, f
    return idx, w1


def _area_tab(n_src, n_dst, scale):
    """computeResizeAreaTab: list of (dst index, src index, float32 weight)."""
    tab = []
    for d in range(n_dst):
        fs1 = d * scale
        fs2 = fs1 + scale
        cell = min(scale, n_src - fs1)
        s1, s2 = math.ceil(fs1), math.floor(fs2)
        s2 = min(s2, n_src - 1)
        s1 = min(s1, s2)
        if s1 - fs1 > 1e-3:
            tab.append((d, s1 - 1, F32((s1 - fs1) / cell)))
        for s in range(s1, s2):
            tab.append((d, s, F32(1.0 / cell)))
        if fs2 - s2 > 1e-3:
            tab.append((d, s2, F32(min(min(fs2 - s2, 1.0), cell) / cell)))
    return tab


def _saturate_u8(v):
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def resize(src: np.ndarray, dsize=None, fx=None, fy=None, interpolation=INTER_LINEAR) -> np.ndarray:
    """One channel (2-D array), dtype float32 / float64 / uint8."""
    src = np.asarray(src)
    assert src.ndim == 2 and src.dtype in (np.float32, np.float64, np.uint8)
    h, w = src.shape
    dw, dh, inv_x, inv_y = dest_size(src.shape, dsize, fx, fy)
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    T = src.dtype.type
    WT = np.float64 if src.dtype == np.float64 else F32
    if interpolation == INTER_NEAREST:
        ifx, ify = 1.0 / inv_x, 1.0 / inv_y
        xs = np.minimum(np.floor(np.arange(dw) * ifx).astype(np.int64), w - 1)
        ys = np.minimum(np.floor(np.arange(dh) * ify).astype(np.int64), h - 1)
        return src[np.ix_(ys, xs)].copy()
    if interpolation == INTER_AREA and scale_x >= 1 and scale_y >= 1:
        isx, isy = cv_round(scale_x), cv_round(scale_y)
        fast = abs(scale_x - isx) < np.finfo(np.float64).eps and abs(scale_y - isy) < np.finfo(np.float64).eps
        out = np.empty((dh, dw), src.dtype)
        if fast:
            scl = F32(1.0 / (isx * isy))
            for dy in range(dh):
                for dx in range(dw):
                    blk = src[dy * isy:(dy + 1) * isy, dx * isx:(dx + 1) * isx]
                    if blk.size < isx * isy:
                        # block sticking out of the image (dsize rounded up): OpenCV averages what is there,
                        # saturate_cast<T>((float)sum / count); nothing there -> 0
                        if blk.size == 0:
                            out[dy, dx] = 0
                            continue
                        acc = (np.int64(0) if src.dtype == np.uint8 else WT(0))
                        for v in blk.ravel():
                            acc = acc + (np.int64(v) if src.dtype == np.uint8 else WT(v))
                            acc = acc if src.dtype == np.uint8 else WT(acc)
                        q = F32(F32(acc) / F32(blk.size))
                        out[dy, dx] = _saturate_u8(q) if src.dtype == np.uint8 else T(q)
                        continue
                    if src.dtype == np.uint8:
                        s = int(blk.astype(np.int64).sum())
                        out[dy, dx] = (s + 2) >> 2 if (isx == 2 and isy == 2) else _saturate_u8(F32(s) * scl)
                    else:
                        acc = WT(0)
                        for v in blk.ravel():
                            acc = WT(acc + WT(v))
                        out[dy, dx] = T(acc * scl) if WT is F32 else T(acc * np.float64(scl))
            return out
        xtab, ytab = _area_tab(w, dw, scale_x), _area_tab(h, dh, scale_y)
        xt = [[(s, a) for (d, s, a) in xtab if d == dx] for dx in range(dw)]
        sums = np.zeros((dh, dw), WT)
        seen = np.zeros(dh, bool)
        for (dy, sy, beta) in ytab:
            row = src[sy].astype(WT)
            buf = np.zeros(dw, WT)
            for dx in range(dw):
                acc = WT(0)
                for (s, a) in xt[dx]:
                    acc = WT(acc + WT(row[s] * WT(a)))
                buf[dx] = acc
            if not seen[dy]:
                sums[dy] = (WT(beta) * buf).astype(WT)
                seen[dy] = True
            else:
                sums[dy] = (sums[dy] + (WT(beta) * buf).astype(WT)).astype(WT)
        return _saturate_u8(sums) if src.dtype == np.uint8 else sums.astype(src.dtype)
    # bilinear (INTER_LINEAR, or INTER_AREA when enlarging)
    area_mode = interpolation == INTER_AREA
    xi, xf = _linear_taps(dw, w, scale_x, inv_x, area_mode)
    yi, yf = _linear_taps(dh, h, scale_y, inv_y, area_mode)
    xi1 = np.minimum(xi + 1, w - 1)
    yi1 = np.minimum(yi + 1, h - 1)
    a0, a1 = (F32(1) - xf).astype(F32), xf
    b0, b1 = (F32(1) - yf).astype(F32), yf
    if src.dtype == np.uint8:
        ia0 = np.clip(np.rint(a0 * F32(2048)), -32768, 32767).astype(np.int64)
        ia1 = np.clip(np.rint(a1 * F32(2048)), -32768, 32767).astype(np.int64)
        ib0 = np.clip(np.rint(b0 * F32(2048)), -32768, 32767).astype(np.int64)
        ib1 = np.clip(np.rint(b1 * F32(2048)), -32768, 32767).astype(np.int64)
        s = src.astype(np.int64)
        rows = s[:, xi] * ia0[None, :] + s[:, xi1] * ia1[None, :]
        r0, r1 = rows[yi], rows[yi1]
        v = (((ib0[:, None] * (r0 >> 4)) >> 16) + ((ib1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
        return np.clip(v, 0, 255).astype(np.uint8)
    s = src.astype(WT)
    rows = ((s[:, xi] * a0[None, :].astype(WT)).astype(WT) + (s[:, xi1] * a1[None, :].astype(WT)).astype(WT)).astype(WT)
    r0, r1 = rows[yi], rows[yi1]
    v = ((r0 * b0[:, None].astype(WT)).astype(WT) + (r1 * b1[:, None].astype(WT)).astype(WT)).astype(WT)
    return v.astype(src.dtype)
