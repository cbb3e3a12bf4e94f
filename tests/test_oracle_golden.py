"""The NumPy oracle against golden vectors produced by the imported reference
(tests/golden/make_golden.py).  float64: bit-exact; float32: bit-exact too,
because the oracle follows the reference's NumPy-2 dtype flow."""
import numpy as np
import pytest

from oracle import tvd_numpy as O
from conftest import split_last_axis


def _eq(a, b):
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
    assert a.shape == b.shape
    if not np.array_equal(a, b):
        raise AssertionError(f"max abs diff {np.abs(a.astype(np.float64) - b).max():.3e}")


CASES = [
    ("g1_aniso2d_f64", dict(mu=0.1, max_num_iter=20)),
    ("g2_aniso2d_f32", dict(mu=0.1, max_num_iter=20)),
    ("g3_aniso_rgb_f32", dict(mu=0.1, max_num_iter=10)),
    ("g3_aniso_rgb_f64", dict(mu=0.1, max_num_iter=10)),
    ("g3_aniso_rgb_odd_f32", dict(mu=0.15, omega=0.7, max_num_iter=7)),
    ("g3_aniso_2ch_odd_f32", dict(mu=0.1, max_num_iter=6)),
    ("g2_aniso2d_odd_f32", dict(mu=0.2, ell=0.3, max_num_iter=9)),
    ("g2_tiny_f32", dict(mu=0.1, max_num_iter=4)),
    ("g2_tiny_col_f32", dict(mu=0.1, max_num_iter=4)),
    ("g2_single_px_f32", dict(mu=0.1, max_num_iter=3)),
    ("g6_jacobi3_f64", dict(mu=0.1, max_num_iter=8, jacobi_maxiter=3)),
    ("g6_jacobi3_rgb_f32", dict(mu=0.1, max_num_iter=5, jacobi_maxiter=3)),
    ("g6_jacobi_tol_f64", dict(mu=0.1, max_num_iter=6, jacobi_maxiter=20, jacobi_tol=1e-3)),
    ("g9_vol3d_f64", dict(mu=0.1, dim=3, max_num_iter=6)),
    ("g9_vol3d_f32", dict(mu=0.1, dim=3, max_num_iter=6)),
]


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_split_bregman_matches_reference(golden, name, kw):
    img = golden[name + "/img"]
    _eq(O.split_bregman_tvd(img, **kw), golden[name + "/out"])


def test_isotropic(golden):
    """The reference's isotropic shrinkage is a numba kernel compiled with
    fastmath=True (split_bregman_tvd.py:107-126); the golden was produced with
    numba stubbed out, i.e. CPython's ``s ** 0.5`` (libm pow) per pixel, which
    can differ from ``sqrt`` in the last bit.  Hence a few-ulp tolerance here
    instead of bit equality."""
    out = O.split_bregman_tvd(golden["g4_iso2d_f64/img"], mu=0.1, max_num_iter=10, isotropic=True)
    np.testing.assert_allclose(out, golden["g4_iso2d_f64/out"], rtol=1e-14, atol=0)


@pytest.mark.parametrize("name,kw,sched", [
    ("g14_adaptive_f64", dict(mu=0.1, max_num_iter=9), lambda it: it % 3 == 1),
    ("g14_adaptive_f32", dict(mu=0.1, max_num_iter=9), lambda it: it % 3 == 1),
    ("g14_adaptive_rgb_f32", dict(mu=0.1, omega=0.9, max_num_iter=6), lambda it: it >= 2),
])
def test_adaptive_regularisation(golden, name, kw, sched):
    _eq(O.split_bregman_tvd(golden[name + "/img"], adaptive=sched, **kw), golden[name + "/out"])


def test_heterogeneous_coefficients(golden):
    g = golden
    _eq(O.split_bregman_tvd(g["g5_het_mu_f64/img"], mu=g["g5_het_mu_f64/mu"], omega=0.1, max_num_iter=12),
        g["g5_het_mu_f64/out"])
    _eq(O.split_bregman_tvd(g["g5_het_mu_f32/img"], mu=g["g5_het_mu_f32/mu"], omega=0.1, max_num_iter=12),
        g["g5_het_mu_f32/out"])
    _eq(O.split_bregman_tvd(g["g5_het_omega_f64/img"], mu=0.1, omega=g["g5_het_omega_f64/omega"], ell=0.25,
                            max_num_iter=8), g["g5_het_omega_f64/out"])
    _eq(O.split_bregman_tvd(g["g5_het_mu_rgb_f32/img"], mu=g["g5_het_mu_rgb_f32/mu"], max_num_iter=6),
        g["g5_het_mu_rgb_f32/out"])


@pytest.mark.parametrize("name,eps", [("g7_eps_f64", 1e-3), ("g7_eps_f32", 1e-3), ("g7_eps_rgb_f32", 2e-3)])
def test_early_stop(golden, name, eps):
    out, _, _, done = O.split_bregman_tvd(golden[name + "/img"], mu=0.1, max_num_iter=100, eps=eps,
                                          return_state=True)
    assert done == int(golden[name + "/iters"])
    assert 1 < done < 100
    _eq(out, golden[name + "/out"])


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_warm_start(golden, tag):
    g = golden
    img = g[f"g8_warm_{tag}/img"]
    u, d, b, _ = O.split_bregman_tvd(img, mu=0.1, max_num_iter=5, return_state=True)
    _eq(u, g[f"g8_warm_{tag}/first"])
    for j, (dj, bj) in enumerate(zip(split_last_axis(g[f"g8_warm_{tag}/d5"]), split_last_axis(g[f"g8_warm_{tag}/b5"]))):
        _eq(d[j], dj)
        _eq(b[j], bj)
    u2, d2, b2, _ = O.split_bregman_tvd(img, mu=0.1, max_num_iter=5, x0=(u, d, b), return_state=True)
    _eq(u2, g[f"g8_warm_{tag}/second"])
    for j in range(2):
        _eq(d2[j], g[f"g8_warm_{tag}/d10"][..., j])
        _eq(b2[j], g[f"g8_warm_{tag}/b10"][..., j])


@pytest.mark.parametrize("tag", ["2d_f64", "2d_f32", "rgb_f32"])
def test_h1(golden, tag):
    _eq(O.h1_regularization(golden[f"g10_h1_{tag}/img"], mu=0.4, omega=1.3, dim=2, jacobi_maxiter=4),
        golden[f"g10_h1_{tag}/out"])


def test_derivatives(golden):
    g = golden
    a2, a3 = g["g11_diff/a2"], g["g11_diff/a3"]
    for ax in range(2):
        _eq(O.backward_diff(a2, ax), g[f"g11_diff/a2_bwd{ax}"])
        _eq(O.forward_diff(a2, ax), g[f"g11_diff/a2_fwd{ax}"])
        _eq(O.backward_diff(a2, ax, h=0.5), g[f"g11_diff/a2_bwd{ax}_h"])
        _eq(O.backward_diff(a3, ax), g[f"g11_diff/a3_bwd{ax}"])
        _eq(O.forward_diff(a3, ax), g[f"g11_diff/a3_fwd{ax}"])
    _eq(O.backward_diff(a3, 2), g["g11_diff/a3_bwd2"])
    _eq(O.forward_diff(a3, 2), g["g11_diff/a3_fwd2"])
    _eq(O.laplace(a2), g["g11_diff/a2_lap"])
    _eq(O.laplace(a2, h=0.5, diffusion_coeff=2.0), g["g11_diff/a2_lap_h"])
    _eq(O.laplace(a3, dim=3), g["g11_diff/a3_lap3"])
    _eq(O.laplace(a2, axis=1), g["g11_diff/a2_lap_ax1"])


def test_config1_512(golden):
    """BASELINE config 1: 512 x 512, 20 iterations (checksums and crops)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import noisy_blocks
    g = golden
    big = noisy_blocks((512, 512), int(g["g12_512/seed"]), np.float64)
    res = O.split_bregman_tvd(big, mu=0.1, max_num_iter=20)
    assert res.sum() == float(g["g12_512/sum"])
    assert (res * res).sum() == float(g["g12_512/sumsq"])
    _eq(res[:16, :16], g["g12_512/crop_tl"])
    _eq(res[-16:, -16:], g["g12_512/crop_br"])
    _eq(res[248:264, 248:264], g["g12_512/crop_mid"])
    res32 = O.split_bregman_tvd(big.astype(np.float32), mu=0.1, max_num_iter=20)
    _eq(res32[248:264, 248:264], g["g12_512/crop_mid_f32"])
    assert res32.astype(np.float64).sum() == float(g["g12_512/sum_f32"])


def test_strict_flow_close_to_numpy2_flow(golden):
    """The pure-float32 flow (what the HIP kernels compute) stays within the
    north-star tolerance of the reference's mixed float32/float64 flow."""
    img = golden["g3_aniso_rgb_f32/img"]
    ref = golden["g3_aniso_rgb_f32/out"]
    out = O.split_bregman_tvd(img, mu=0.1, max_num_iter=10, promotion="strict")
    assert out.dtype == np.float32
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-6)
    assert not np.array_equal(out, ref)  # the flows are genuinely different
