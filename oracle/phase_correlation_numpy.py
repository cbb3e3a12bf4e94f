"""CPU statement of the patch translation estimator of darsia_amd's registration: FFT phase correlation with a parabolic
sub-pixel fit.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE (same rule as oracle/tvd_numpy.py).  This estimator has NO counterpart in the reference:
DarSIA's ``TranslationEstimator`` (image registration by ORB features + RANSAC homography, multi_image_analysis/
translationestimator.py) is OpenCV through and through and out of this repository's reach; the registration
path takes any object with its ``find_effective_translation`` signature, and this is the one darsia_amd ships.  So there is nothing
to pin against: the file states the algorithm the device kernels (darsia_amd/csrc/dxh_dense.hip, ``dxh_phase_correlation``)
implement, in NumPy's float64, and tests/test_phasecorr_gpu.py compares the two to a tolerance (different FFT algorithms).

Per patch pair (a = source patch, b = destination patch, same pixel rectangle of both images):
  * colour patches are averaged over their channels; everything is float64
  * a - mean(a) and b - mean(b) are multiplied by the separable Hann window of the PATCH size (np.hanning)
  * both are zero-padded to (P0, P1), powers of two >= the largest patch of the call (at least 8), and transformed
    (patches longer than 8192 pixels are first reduced to cells of dec x dec pixels, see phase_correlation_patches)
  * cross = F(b) conj(F(a)), divided by max(|cross|, 1e-12); corr = real part of the inverse transform
  * the peak of corr (first maximum in row-major order) is refined by a parabola through its circular neighbours per axis,
    unwrapped to [-P/2, P/2), and rejected when it exceeds ``max_shift_fraction`` of the patch size in either direction,
    when it is not finite, or when the peak itself stays below ``min_peak_ratio * sqrt(ln N / N)``, N = P0 P1: that square root
    is the level the maximum of the surface reaches for two UNRELATED patterns (each of the N values is the mean of N unit
    phases, standard deviation 1 / sqrt(2 N)); a flat patch has a zero transform and a surface of zeros, two equal patterns a
    peak of 1.  (The reference's ORB + RANSAC estimator reports such patches as not intact and the spline interpolates over
    them, translationanalysis.py:225-275.)
"""
import numpy as np


def next_pow2(n: int) -> int:
    p = 8
    while p < n:
        p *= 2
    return p


def phase_correlation(a, b, padded_shape=None, max_shift_fraction: float = 0.25, min_peak_ratio: float = 4.0, return_peak=False):
    """(dx, dy) of patch ``b`` relative to patch ``a`` and whether the estimate is accepted; ``padded_shape`` = (P0, P1)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.ndim == 3:
        a, b = a.mean(axis=2), b.mean(axis=2)
    if a.shape != b.shape or min(a.shape) < 8:
        return None, False
    ph, pw = a.shape
    p0, p1 = (next_pow2(ph), next_pow2(pw)) if padded_shape is None else padded_shape
    wy, wx = np.hanning(ph)[:, None], np.hanning(pw)[None, :]
    fa = np.fft.fft2((a - a.mean()) * wy * wx, s=(p0, p1))
    fb = np.fft.fft2((b - b.mean()) * wy * wx, s=(p0, p1))
    cross = fb * np.conj(fa)
    cross /= np.maximum(np.abs(cross), 1e-12)
    corr = np.fft.ifft2(cross).real
    py, px = np.unravel_index(np.argmax(corr), corr.shape)

    def refine(c_m, c_0, c_p):
        den = c_m - 2 * c_0 + c_p
        return 0.0 if den == 0 else 0.5 * (c_m - c_p) / den

    dy = py + refine(corr[(py - 1) % p0, px], corr[py, px], corr[(py + 1) % p0, px])
    dx = px + refine(corr[py, (px - 1) % p1], corr[py, px], corr[py, (px + 1) % p1])
    if dy > p0 / 2:
        dy -= p0
    if dx > p1 / 2:
        dx -= p1
    peak = float(corr[py, px])
    n = float(p0) * float(p1)
    min_peak = min_peak_ratio * np.sqrt(np.log(n) / n) if min_peak_ratio > 0 else -np.inf
    ok = bool(abs(dy) <= max_shift_fraction * ph and abs(dx) <= max_shift_fraction * pw and np.isfinite(peak) and peak >= min_peak)
    return ((dx, dy), ok, peak) if return_peak else ((dx, dy), ok)


MAX_TRANSFORM = 8192   # longest transform of the device implementation (128 KB of LDS)


def cells(a, dec: int):
    """Patch read in cells of dec x dec pixels (their mean, channels averaged first); rows / columns that do not fill a cell drop."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 3:
        a = a.mean(axis=2)
    if dec == 1:
        return a
    ph, pw = a.shape[0] // dec, a.shape[1] // dec
    return a[:ph * dec, :pw * dec].reshape(ph, dec, pw, dec).mean(axis=(1, 3))


def phase_correlation_patches(img_src, img_dst, rois, max_shift_fraction: float = 0.25, min_peak_ratio: float = 4.0):
    """All patches of a call: ``rois`` = (n, 4) rows r0, r1, c0, c1; one padded shape for the whole call.  When the largest patch is
    longer than MAX_TRANSFORM the patches are read in cells of dec x dec pixels (dec the smallest power of two that fits) and the
    shifts are scaled back to pixels."""
    rois = np.asarray(rois, dtype=np.int64).reshape(-1, 4)
    maxh, maxw = int((rois[:, 1] - rois[:, 0]).max()), int((rois[:, 3] - rois[:, 2]).max())
    dec = 1
    while maxh // dec > MAX_TRANSFORM or maxw // dec > MAX_TRANSFORM:
        dec *= 2
    p0, p1 = next_pow2(maxh // dec), next_pow2(maxw // dec)
    out = []
    for r0, r1, c0, c1 in rois:
        shift, ok = phase_correlation(cells(img_src[r0:r1, c0:c1], dec), cells(img_dst[r0:r1, c0:c1], dec), (p0, p1), max_shift_fraction,
                                      min_peak_ratio)
        out.append((None if shift is None else (shift[0] * dec, shift[1] * dec), ok))
    return out
