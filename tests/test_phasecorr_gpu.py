"""The device patch estimator (dxh_phase_correlation) against its NumPy statement (oracle/phase_correlation_numpy.py): same
algorithm, different FFTs, so shifts agree to a tolerance (1e-7 px in float64); acceptance flags and peak positions exactly."""
import numpy as np
import pytest

import darsia_amd as da
from darsia_amd.imageregistration import PhaseCorrelationEstimator
from oracle import phase_correlation_numpy as P

pytestmark = pytest.mark.gpu


def _pair(h, w, seed, shift=(3.3, -5.6), channels=1):
    from scipy.ndimage import gaussian_filter, shift as nd_shift

    rng = np.random.default_rng(seed)
    tex = np.stack([gaussian_filter(rng.random((h, w)), 2.0) for _ in range(channels)], axis=-1)
    moved = np.stack([nd_shift(tex[..., c], shift, order=3, mode="nearest") for c in range(channels)], axis=-1)
    return (tex[..., 0], moved[..., 0]) if channels == 1 else (tex, moved)


def _compare(a, b, rois, frac=0.25, tol=1e-7):
    got = PhaseCorrelationEstimator(frac).find_effective_translations(a, b, rois)
    want = P.phase_correlation_patches(a, b, rois, frac)
    assert len(got) == len(want)
    for (t, ok), (w_shift, w_ok) in zip(got, want):
        assert ok == w_ok
        if ok:
            assert t.shape == (2, 3) and np.array_equal(t[:, :2], np.eye(2))
            np.testing.assert_allclose(t[:, 2], w_shift, rtol=0, atol=tol)
    return got


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.uint8, np.uint16])
@pytest.mark.parametrize("channels", [1, 3])
def test_patches_of_a_pair_match_the_numpy_statement(dtype, channels):
    a, b = _pair(150, 230, 3, channels=channels)
    if np.dtype(dtype).kind == "u":
        a, b = (a * np.iinfo(dtype).max).astype(dtype), (b * np.iinfo(dtype).max).astype(dtype)
    else:
        a, b = a.astype(dtype), b.astype(dtype)
    # ragged rectangles: different sizes in one call (one padded shape for all), one too small to be estimated
    rois = [(0, 70, 0, 110), (60, 150, 100, 230), (10, 137, 5, 201), (0, 150, 0, 230), (20, 27, 30, 120), (40, 104, 50, 178)]
    got = _compare(a, b, rois)
    assert [ok for _, ok in got] == [True, True, True, True, False, True]
    assert abs(got[3][0][0, 2] + 5.6) < 0.5 and abs(got[3][0][1, 2] - 3.3) < 0.5     # and it is the shift that was applied


def test_rejections_masks_and_the_single_patch_signature():
    a, b = _pair(96, 128, 5, shift=(30.0, 2.0))
    est = PhaseCorrelationEstimator(0.25)
    # a 30-pixel shift on a 96-pixel patch exceeds a quarter of the patch: rejected, as the statement says
    assert _compare(a, b, [(0, 96, 0, 128)])[0][1] is False
    assert _compare(a, b, [(0, 96, 0, 128)], frac=0.5)[0][1] is True
    t, ok = est.find_effective_translation(a[:, :100], b[:, :100])
    want, w_ok = P.phase_correlation(a[:, :100], b[:, :100])
    assert ok == w_ok
    empty, full = np.zeros(a.shape, bool), np.ones(a.shape, bool)
    assert PhaseCorrelationEstimator(0.5).find_effective_translations(a, b, [(0, 96, 0, 128)], empty, full)[0] == (None, False)
    assert PhaseCorrelationEstimator(0.5).find_effective_translations(a, b, [(0, 96, 0, 128)], full, full)[0][1] is True
    assert est.find_effective_translation(a, b[:, :100]) == (None, False)          # shapes differ


def test_large_patches_and_many_patches():
    """Transforms up to 8192 points (128 KB of LDS per row); the 10 x 20 patch grid of configs[3] at a quarter of its size."""
    a, b = _pair(40, 5000, 7, shift=(1.2, 11.5))
    _compare(a, b, [(0, 40, 0, 5000)], tol=1e-6)
    a, b = _pair(1000, 2000, 8, shift=(2.4, -3.1))
    patches = da.Patches(da.Image(a), [10, 20], rel_overlap=0.1)
    rois = [patches.rois[i][j] for i in range(10) for j in range(20)]
    got = PhaseCorrelationEstimator().find_effective_translations(a, b, rois)
    want = P.phase_correlation_patches(a, b, [(r[0].start, r[0].stop, r[1].start, r[1].stop) for r in rois])
    assert [ok for _, ok in got] == [ok for _, ok in want]
    np.testing.assert_allclose([t[:, 2] for t, ok in got if ok], [s for s, ok in want if ok], rtol=0, atol=1e-7)
    # longer than the longest transform: read in 2 x 2 cells, shifts back in pixels
    a, b = _pair(36, 9000, 10, shift=(2.0, -14.0))
    got = _compare(a, b, [(0, 36, 0, 9000), (3, 30, 100, 8800)], tol=1e-6)
    assert abs(got[0][0][0, 2] + 14.0) < 1.5 and abs(got[0][0][1, 2] - 2.0) < 1.5


def test_patches_in_several_passes_give_the_same_shifts(monkeypatch):
    """When the two complex planes per patch do not fit the memory budget the patches go through in chunks (DXH_PC_CHUNK forces it)."""
    a, b = _pair(200, 300, 9, shift=(-2.2, 4.4))
    rois = [(r, r + 64, c, c + 96) for r in (0, 60, 136) for c in (0, 100, 204)]
    est = PhaseCorrelationEstimator()
    whole = est.find_effective_translations(a, b, rois)
    monkeypatch.setenv("DXH_PC_CHUNK", "4")
    chunked = est.find_effective_translations(a, b, rois)
    assert [ok for _, ok in whole] == [ok for _, ok in chunked] and all(ok for _, ok in whole)
    assert all(np.array_equal(t0, t1) for (t0, _), (t1, _) in zip(whole, chunked))


def test_flat_noise_only_nan_and_mostly_masked_patches_are_not_intact():
    """What the reference's ORB + RANSAC matcher reports as not intact (translationanalysis.py:225-275) must not become a
    control point of the spline here either: a flat patch, two unrelated noise patches, a patch holding a NaN, a patch
    most of which is masked out.  Flags equal the NumPy statement's; the peak values agree."""
    rng = np.random.default_rng(12)
    h, w = 128, 160
    a, b = _pair(h, 4 * w, 9, shift=(2.5, -3.25))
    a, b = a.copy(), b.copy()
    a[:, w:2 * w] = b[:, w:2 * w] = 0.37                      # patch 1: flat in both frames
    a[:, 2 * w:3 * w] = rng.random((h, w))                      # patch 2: unrelated noise
    b[:, 2 * w:3 * w] = rng.random((h, w))
    rois = [(0, h, k * w, (k + 1) * w) for k in range(4)]
    est = PhaseCorrelationEstimator(0.25)
    got = est.find_effective_translations(a, b, rois)
    want = P.phase_correlation_patches(a, b, rois, 0.25)
    assert [ok for _, ok in got] == [ok for _, ok in want] == [True, False, False, True]
    peaks = [P.phase_correlation(a[r0:r1, c0:c1], b[r0:r1, c0:c1], (128, 256), return_peak=True)[2] for r0, r1, c0, c1 in rois]
    np.testing.assert_allclose(est.last_peaks, peaks, rtol=0, atol=1e-9)
    assert abs(est.last_peaks[1]) < 1e-12 and est.last_peaks[2] < 0.05 < est.last_peaks[0]
    # the test can be switched off: then the noise patch is whatever its random peak says
    assert PhaseCorrelationEstimator(0.5, min_peak_ratio=0.0).find_effective_translations(a, b, rois[:1])[0][1] is True
    # a NaN anywhere in a patch poisons its transform: not intact (and the others are unaffected)
    a_nan = a.copy()
    a_nan[5, 3 * w + 7] = np.nan
    flags = [ok for _, ok in est.find_effective_translations(a_nan, b, rois)]
    assert flags == [True, False, False, False]
    # masks: a patch counts only if at least min_active_fraction of it is active in both masks
    m_src, m_dst = np.ones(a.shape, bool), np.ones(a.shape, bool)
    m_src[:, :w][:, : int(0.7 * w)] = False                    # patch 0: 30 % active in the source mask
    m_dst[: h // 4, 3 * w:] = False                             # patch 3: 75 % active in the destination mask
    flags = [ok for _, ok in est.find_effective_translations(a, b, rois, m_src, m_dst)]
    assert flags == [False, False, False, True]
    assert [ok for _, ok in PhaseCorrelationEstimator(0.25, min_active_fraction=0.2).find_effective_translations(a, b, rois, m_src, m_dst)] == \
        [True, False, False, True]
