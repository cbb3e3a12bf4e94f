"""CPU checks of the registration oracle (parity with OpenCV is UNPINNED: cv2 is not installed and the
reference's own registration test needs absent image files).  What can be pinned here: exact properties of
the restated algorithm, and the thin-plate-spline evaluation against SciPy itself."""
import numpy as np
from scipy.interpolate import RBFInterpolator

from oracle import warp_numpy as W


def test_identity_and_integer_translation_are_exact():
    rng = np.random.default_rng(0)
    img = rng.random((20, 30)).astype(np.float32)
    src = np.array([[0, 0], [29, 0], [29, 19], [0, 19]], np.float32)
    m = W.get_perspective_transform(src, src)
    np.testing.assert_allclose(m, np.eye(3), atol=1e-12)
    assert np.array_equal(W.warp_perspective(img, m, 30, 20), img)
    out = W.warp_perspective(img, W.get_perspective_transform(src, src + np.float32([2, 1])), 30, 20)
    assert np.array_equal(out[1:, 2:], img[:-1, :-2]) and out[0].max() == 0 and out[:, :2].max() == 0


def test_subpixel_shift_uses_one_32nd_pixel_bilinear_weights():
    rng = np.random.default_rng(0)
    img = rng.random((20, 30)).astype(np.float32)
    src = np.array([[0, 0], [29, 0], [29, 19], [0, 19]], np.float32)
    out = W.warp_perspective(img, W.get_perspective_transform(src, src + np.float32([0.5, 0.25])), 30, 20)
    expect = (np.float32(0.25) * np.float32(0.5) * img[4, 4] + np.float32(0.25) * np.float32(0.5) * img[4, 5]
              + np.float32(0.75) * np.float32(0.5) * img[5, 4] + np.float32(0.75) * np.float32(0.5) * img[5, 5])
    assert out[5, 5] == expect


def test_homography_maps_the_four_points():
    rng = np.random.default_rng(1)
    src = np.array([[0, 0], [40, 0], [40, 30], [0, 30]], np.float32)
    dst = src + rng.normal(0, 2, src.shape).astype(np.float32)
    m = W.get_perspective_transform(src, dst)
    p = np.c_[src, np.ones(4)] @ m.T
    np.testing.assert_allclose(p[:, :2] / p[:, 2:], dst, atol=1e-9)


def test_thin_plate_spline_matches_scipy():
    rng = np.random.default_rng(2)
    c = rng.random((25, 2)) * 100
    v = rng.normal(size=(25, 2))
    r = RBFInterpolator(c, v)
    pts = rng.random((200, 2)) * 120 - 10
    mine = W.thin_plate_spline_eval(pts, r.y, r._coeffs, r._shift, r._scale, r.powers)
    np.testing.assert_allclose(mine, r(pts), rtol=1e-11, atol=1e-11)
