"""The oracle's same-dtype ("strict") flow against vectors produced by EXECUTING the reference under NumPy 1.26
(tests/golden/make_golden_np1.py): under value-based casting ``float32_array / np.float64_scalar`` stays float32
(utils/linear_solvers/jacobi.py:101-114), so the reference's own float32 arithmetic is the strict flow -- bit for bit.
This is the pin of the flow ``bench.py``'s headline times (``promotion="native"``, tests/test_np1_golden_gpu.py)."""
import os
import sys

import numpy as np
import pytest

from oracle import tvd_numpy as O
from conftest import split_last_axis

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def np1():
    return np.load(os.path.join(ROOT, "tests", "golden", "tvd_np1_golden.npz"), allow_pickle=False)


def _eq(a, b):
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
    assert a.shape == b.shape
    if not np.array_equal(a, b):
        raise AssertionError(f"max abs diff {np.abs(a.astype(np.float64) - b).max():.3e}")


CASES = [
    ("p2_aniso2d_f32", dict(mu=0.1, max_num_iter=20)),
    ("p3_aniso_rgb_f32", dict(mu=0.1, max_num_iter=10)),
    ("p3_aniso_rgb_odd_f32", dict(mu=0.15, omega=0.7, max_num_iter=7)),
    ("p3_aniso_2ch_odd_f32", dict(mu=0.1, max_num_iter=6)),
    ("p2_aniso2d_odd_f32", dict(mu=0.2, ell=0.3, max_num_iter=9)),
    ("p2_tiny_f32", dict(mu=0.1, max_num_iter=4)),
    ("p2_tiny_col_f32", dict(mu=0.1, max_num_iter=4)),
    ("p20_mu1_rgb_f32", dict(mu=1.0, max_num_iter=60)),
    ("p21_ell1_f32", dict(mu=0.1, ell=1.0, max_num_iter=100)),
    ("p22_mu1_gray_f32", dict(mu=1.0, omega=0.5, max_num_iter=40)),
    ("p6_jacobi3_rgb_f32", dict(mu=0.1, max_num_iter=5, jacobi_maxiter=3)),
    ("p9_vol3d_f32", dict(mu=0.1, dim=3, max_num_iter=6)),
]


def test_generated_under_numpy_1(np1):
    assert str(np1["meta/numpy"]).startswith("1.")
    assert str(np1["meta/p10_jacobi_dtype"]) == "float32"   # the reference's Jacobi keeps float32 there


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_strict_flow_is_the_reference_under_numpy_1(np1, name, kw):
    img = np1[name + "/img"]
    assert img.dtype == np.float32
    _eq(O.split_bregman_tvd(img, promotion="strict", **kw), np1[name + "/out"])


def test_the_two_flows_part_where_the_shrinkage_is_active(np1, golden):
    """Same input, the two NumPy generations of the reference: equal within the north-star tolerance in the benign regime,
    visibly apart once elements sit near the threshold -- the reason both flows exist on the device."""
    a, b = np1["p2_aniso2d_f32/out"], golden["g2_aniso2d_f32/out"]
    assert not np.array_equal(a, b)
    np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-6)
    far = O.split_bregman_tvd(np1["p21_ell1_f32/img"], mu=0.1, ell=1.0, max_num_iter=100)   # NumPy-2 flow
    d = np.abs(far.astype(np.float64) - np1["p21_ell1_f32/out"])
    assert d.max() > 1e-5   # > 10 x the tolerance's absolute part


@pytest.mark.parametrize("name,eps", [("p7_eps_f32", 1e-3), ("p7_eps_rgb_f32", 2e-3)])
def test_early_stop(np1, name, eps):
    out, _, _, done = O.split_bregman_tvd(np1[name + "/img"], mu=0.1, max_num_iter=100, eps=eps, promotion="strict",
                                          return_state=True)
    assert done == int(np1[name + "/iters"])
    _eq(out, np1[name + "/out"])


def test_float32_coefficient_arrays(np1):
    _eq(O.split_bregman_tvd(np1["p5_het_mu_f32/img"], mu=np1["p5_het_mu_f32/mu"], omega=0.1, max_num_iter=12,
                            promotion="strict"), np1["p5_het_mu_f32/out"])
    _eq(O.split_bregman_tvd(np1["p5_het_mu_rgb_f32/img"], mu=np1["p5_het_mu_rgb_f32/mu"], max_num_iter=6,
                            promotion="strict"), np1["p5_het_mu_rgb_f32/out"])


def test_warm_start(np1):
    img = np1["p8_warm_f32/img"]
    z = [np.zeros(img.shape, np.float32) for _ in range(2)]
    u5, d5, b5, _ = O.split_bregman_tvd(img, mu=0.1, max_num_iter=5, x0=(img.copy(), z, [a.copy() for a in z]),
                                        promotion="strict", return_state=True)
    _eq(u5, np1["p8_warm_f32/first"])
    for j in range(2):
        _eq(d5[j], split_last_axis(np1["p8_warm_f32/d5"])[j])
        _eq(b5[j], split_last_axis(np1["p8_warm_f32/b5"])[j])
    u10, d10, b10, _ = O.split_bregman_tvd(img, mu=0.1, max_num_iter=5, x0=(u5, d5, b5), promotion="strict",
                                           return_state=True)
    _eq(u10, np1["p8_warm_f32/second"])
    for j in range(2):
        _eq(d10[j], split_last_axis(np1["p8_warm_f32/d10"])[j])
        _eq(b10[j], split_last_axis(np1["p8_warm_f32/b10"])[j])


def test_config1_512_float32(np1):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import noisy_blocks  # pure NumPy helper (nothing of the reference is touched at import)
    big32 = noisy_blocks((512, 512), int(np1["p12_512/seed"]), np.float64).astype(np.float32)
    res = O.split_bregman_tvd(big32, mu=0.1, max_num_iter=20, promotion="strict")
    _eq(res[:16, :16], np1["p12_512/crop_tl_f32"])
    _eq(res[248:264, 248:264], np1["p12_512/crop_mid_f32"])
    _eq(res[-16:, -16:], np1["p12_512/crop_br_f32"])
    assert float(res.astype(np.float64).sum()) == float(np1["p12_512/sum_f32"])


def test_jacobi_and_h1(np1):
    img = np1["p10_jacobi4_f32/img"]
    _eq(O.jacobi(img, np.float32(1.3) * img, 1.3, 0.4, dim=2, maxiter=4, promotion="strict").astype(np.float32, copy=False),
        np1["p10_jacobi4_f32/out"])
    rgb = np1["p10_h1_rgb_f32/img"]
    _eq(O.h1_regularization(rgb, mu=0.4, omega=1.3, dim=2, jacobi_maxiter=4, promotion="strict"), np1["p10_h1_rgb_f32/out"])
