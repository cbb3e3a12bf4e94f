"""The NumPy restatement of the reference's MG / CG solvers against vectors produced by the reference itself
(tests/golden/make_golden_solvers.py, NumPy 1.26.4 / SciPy 1.7.1)."""
import os

import numpy as np
import pytest

from oracle import solvers_numpy as so
from oracle import tvd_numpy as tv

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "solvers_golden.npz"))


def g(name, key):
    return G[f"{name}/{key}"]


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_mg_2d_bit_exact(tag):
    n = f"mg_2d_{tag}"
    s = so.MG(depth=2, smoother_iterations=3, maxiter=2, mass_coeff=1.0, diffusion_coeff=0.5, dim=2)
    out = s(g(n, "x0"), g(n, "rhs"))
    assert out.dtype == g(n, "out").dtype
    np.testing.assert_array_equal(out, g(n, "out"))


def test_mg_depth0_tol_3d_bit_exact():
    out = so.MG(depth=0, smoother_iterations=2, maxiter=1, mass_coeff=0.3, diffusion_coeff=1.5, dim=2)(g("mg_d0", "x0"), g("mg_d0", "rhs"))
    np.testing.assert_array_equal(out, g("mg_d0", "out"))
    out = so.MG(depth=1, smoother_iterations=4, maxiter=50, tol=1e-3, mass_coeff=1.0, diffusion_coeff=0.2, dim=2)(
        g("mg_tol", "x0"), g("mg_tol", "rhs"))
    np.testing.assert_array_equal(out, g("mg_tol", "out"))
    out = so.MG(depth=1, smoother_iterations=2, maxiter=2, mass_coeff=1.0, diffusion_coeff=0.7, dim=3)(g("mg_3d", "x0"), g("mg_3d", "rhs"))
    np.testing.assert_array_equal(out, g("mg_3d", "out"))


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_mg_heterogeneous_two_calls_bit_exact(tag):
    n = f"mg_het_{tag}"
    s = so.MG(depth=1, smoother_iterations=2, maxiter=2, mass_coeff=g(n, "mass").copy(), diffusion_coeff=g(n, "diff").copy(), dim=2)
    o1 = s(g(n, "x0"), g(n, "rhs"))
    o2 = s(o1.copy(), g(n, "rhs"))
    np.testing.assert_array_equal(o1, g(n, "out1"))
    np.testing.assert_array_equal(o2, g(n, "out2"))
    # the coefficient arrays of the object were smeared by the cycles, exactly as in the reference
    np.testing.assert_array_equal(s.mass_coeff, g(n, "mass_after"))
    np.testing.assert_array_equal(s.diffusion_coeff, g(n, "diff_after"))
    assert not np.array_equal(s.mass_coeff, g(n, "mass"))


def test_mg_channel_arrays_raise_like_the_reference():
    assert str(g("mg_channels", "raises")) == "ValueError"
    with pytest.raises(ValueError):
        so.MG(depth=1, smoother_iterations=1, maxiter=1, mass_coeff=1.0, diffusion_coeff=1.0, dim=2)(
            np.ones((8, 8, 3)), np.ones((8, 8, 3)))


@pytest.mark.parametrize("name,kw", [
    ("cg_2d", dict(mass_coeff=1.0, diffusion_coeff=0.5, dim=2)),
    ("cg_het", dict(mass_coeff=0.7, diffusion_coeff="diff", dim=2)),
    ("cg_3d", dict(mass_coeff=1.0, diffusion_coeff=0.5, dim=3)),
])
def test_cg_to_solver_tolerance(name, kw):
    kw = dict(kw)
    if kw["diffusion_coeff"] == "diff":
        kw["diffusion_coeff"] = g(name, "diff")
    s = so.CG(maxiter=500, tol=1e-10, **kw)
    out = s(g(name, "x0"), g(name, "rhs"))
    # both solved the same SPD system to |r| <= 1e-10 |b|: the solutions agree far below 1e-8
    np.testing.assert_allclose(out, g(name, "out"), rtol=0, atol=1e-8)
    res = g(name, "rhs") - s.apply(g(name, "out"))
    assert np.linalg.norm(res) <= 1e-9 * np.linalg.norm(g(name, "rhs"))


def test_cg_few_iterations_follow_the_same_recurrence():
    s = so.CG(maxiter=3, tol=1e-12, mass_coeff=1.0, diffusion_coeff=0.5, dim=2)
    out = s(g("cg_few", "x0"), g("cg_few", "rhs"))
    np.testing.assert_allclose(out, g("cg_few", "out"), rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("name,dim,kw,solver", [
    ("tvd_mg_f64", 2, dict(mu=0.1, max_num_iter=6), dict(depth=1, smoother_iterations=2, maxiter=1)),
    ("tvd_mg_f32", 2, dict(mu=0.1, max_num_iter=6), dict(depth=1, smoother_iterations=2, maxiter=1)),
    ("tvd_mg_3d", 3, dict(mu=0.05, omega=0.8, max_num_iter=4), dict(depth=1, smoother_iterations=2, maxiter=2)),
])
def test_tvd_with_mg_bit_exact(name, dim, kw, solver):
    img = g(name, "img")
    out = tv.split_bregman_tvd(img.copy(), dim=dim, promotion="strict", solver=so.MG(**solver), **kw)
    assert out.dtype == img.dtype
    np.testing.assert_array_equal(out, g(name, "out"))


def test_tvd_with_cg_to_tolerance():
    img = g("tvd_cg_f64", "img")
    out = tv.split_bregman_tvd(img.copy(), mu=0.1, max_num_iter=5, promotion="strict", solver=so.CG(maxiter=200, tol=1e-10))
    np.testing.assert_allclose(out, g("tvd_cg_f64", "out"), rtol=0, atol=1e-7)
