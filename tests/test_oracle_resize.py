"""Properties of the cv2.resize restatement (oracle/resize_numpy.py).  OpenCV itself is not available here:
parity with cv2 is unpinned; these tests pin what the published algorithm implies."""
import numpy as np
import pytest

from oracle import resize_numpy as R


def test_identity_and_integer_area_and_nearest():
    rng = np.random.default_rng(0)
    a = rng.random((12, 16)).astype(np.float32)
    for interp in (R.INTER_LINEAR, R.INTER_AREA, R.INTER_NEAREST):
        np.testing.assert_array_equal(R.resize(a, fx=1.0, fy=1.0, interpolation=interp), a)
    blk = R.resize(a, dsize=(4, 3), interpolation=R.INTER_AREA)  # 4 x 4 blocks
    np.testing.assert_allclose(blk, a.reshape(3, 4, 4, 4).mean((1, 3)), rtol=3e-7)
    nn = R.resize(a, dsize=(8, 6), interpolation=R.INTER_NEAREST)
    np.testing.assert_array_equal(nn, a[::2, ::2])
    u = (a * 255).astype(np.uint8)
    np.testing.assert_array_equal(R.resize(u, dsize=(8, 6), interpolation=R.INTER_AREA),
                                  ((u.astype(int).reshape(6, 2, 8, 2).sum((1, 3)) + 2) >> 2).astype(np.uint8))


def test_linear_taps_follow_pixel_centres():
    a = np.arange(8, dtype=np.float64)[None, :].repeat(2, 0)
    up = R.resize(a, dsize=(16, 2))  # centres at (dx + 0.5) / 2 - 0.5
    expect = np.clip((np.arange(16) + 0.5) / 2 - 0.5, 0, 7)
    np.testing.assert_allclose(up[0], expect, atol=1e-6)


def test_fractional_area_preserves_the_mean_and_dest_size_rounds_half_to_even():
    rng = np.random.default_rng(1)
    a = rng.random((20, 30))
    out = R.resize(a, dsize=(7, 9), interpolation=R.INTER_AREA)
    assert out.shape == (9, 7) and abs(out.mean() - a.mean()) < 1e-6
    assert R.dest_size((5, 5), None, 0.5, 0.5)[:2] == (2, 2) and R.dest_size((7, 7), None, 0.5, 0.5)[:2] == (4, 4)
    odd = R.resize(a[:7, :7].astype(np.float32), fx=0.5, fy=0.5, interpolation=R.INTER_AREA)  # last block sticks out
    assert odd.shape == (4, 4) and odd[3, 3] == a[6, 6].astype(np.float32)
