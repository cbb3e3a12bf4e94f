"""MG / CG on the device (through the C ABI) against the reference's own outputs (tests/golden/solvers_golden.npz)
and against the NumPy oracle on other shapes.  MG: bit-exact.  CG: to the solver tolerance."""
import os

import numpy as np
import pytest

import darsia_amd as da
from oracle import solvers_numpy as so
from oracle import tvd_numpy as tv

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "solvers_golden.npz"))


def g(name, key):
    return G[f"{name}/{key}"]


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_mg_2d_golden_bit_exact(tag):
    n = f"mg_2d_{tag}"
    s = da.MG(depth=2, smoother_iterations=3, maxiter=2, mass_coeff=1.0, diffusion_coeff=0.5, dim=2)
    out = s(g(n, "x0"), g(n, "rhs"))
    assert out.dtype == g(n, "out").dtype
    np.testing.assert_array_equal(out, g(n, "out"))


def test_mg_depth0_tol_3d_golden_bit_exact():
    out = da.MG(depth=0, smoother_iterations=2, maxiter=1, mass_coeff=0.3, diffusion_coeff=1.5, dim=2)(g("mg_d0", "x0"), g("mg_d0", "rhs"))
    np.testing.assert_array_equal(out, g("mg_d0", "out"))
    out = da.MG(depth=1, smoother_iterations=4, maxiter=50, tol=1e-3, mass_coeff=1.0, diffusion_coeff=0.2, dim=2)(
        g("mg_tol", "x0"), g("mg_tol", "rhs"))
    np.testing.assert_array_equal(out, g("mg_tol", "out"))
    out = da.MG(depth=1, smoother_iterations=2, maxiter=2, mass_coeff=1.0, diffusion_coeff=0.7, dim=3)(g("mg_3d", "x0"), g("mg_3d", "rhs"))
    np.testing.assert_array_equal(out, g("mg_3d", "out"))


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_mg_heterogeneous_two_calls_golden_bit_exact(tag):
    n = f"mg_het_{tag}"
    s = da.MG(depth=1, smoother_iterations=2, maxiter=2, mass_coeff=g(n, "mass").copy(), diffusion_coeff=g(n, "diff").copy(), dim=2)
    o1 = s(g(n, "x0"), g(n, "rhs"))
    o2 = s(o1.copy(), g(n, "rhs"))
    np.testing.assert_array_equal(o1, g(n, "out1"))
    np.testing.assert_array_equal(o2, g(n, "out2"))
    np.testing.assert_array_equal(s.mass_coeff, g(n, "mass_after"))
    np.testing.assert_array_equal(s.diffusion_coeff, g(n, "diff_after"))


def test_mg_rejects_channel_arrays_and_bad_depth():
    with pytest.raises(ValueError):
        da.MG(depth=1, smoother_iterations=1, maxiter=1, mass_coeff=1.0, diffusion_coeff=1.0, dim=2)(
            np.ones((8, 8, 3)), np.ones((8, 8, 3)))
    with pytest.raises(ValueError):  # 5 -> 2 -> 1 -> 0 rows
        da.MG(depth=3, smoother_iterations=1, maxiter=1, mass_coeff=1.0, diffusion_coeff=1.0, dim=2)(
            np.ones((5, 64)), np.ones((5, 64)))
    with pytest.raises(ValueError):  # scalar solver, then an array coefficient
        s = da.MG(depth=1, smoother_iterations=1, maxiter=1, mass_coeff=1.0, diffusion_coeff=1.0, dim=2)
        s.update_params(diffusion_coeff=np.ones((16, 16)))
        s(np.ones((16, 16)), np.ones((16, 16)))


@pytest.mark.parametrize("shape,dim,dtype,depth", [((257, 391), 2, np.float32, 3), ((130, 96), 2, np.float64, 2),
                                                   ((33, 47, 29), 3, np.float32, 2), ((64, 40, 18), 3, np.float64, 1)])
def test_mg_vs_oracle_other_shapes_bit_exact(shape, dim, dtype, depth):
    rng = np.random.default_rng(3)
    x0 = rng.random(shape).astype(dtype)
    rhs = rng.random(shape).astype(dtype)
    kw = dict(depth=depth, smoother_iterations=3, maxiter=3, mass_coeff=0.9, diffusion_coeff=0.35, dim=dim)
    np.testing.assert_array_equal(da.MG(**kw)(x0, rhs), so.MG(**kw)(x0, rhs))


def test_mg_heterogeneous_3d_vs_oracle_bit_exact():
    rng = np.random.default_rng(4)
    shape = (21, 30, 17)
    x0, rhs = rng.random(shape), rng.random(shape)
    diff = 0.2 + rng.random(shape)
    a = da.MG(depth=1, smoother_iterations=2, maxiter=2, mass_coeff=0.6, diffusion_coeff=diff.copy(), dim=3)
    b = so.MG(depth=1, smoother_iterations=2, maxiter=2, mass_coeff=0.6, diffusion_coeff=diff.copy(), dim=3)
    np.testing.assert_array_equal(a(x0, rhs), b(x0, rhs))
    np.testing.assert_array_equal(a.diffusion_coeff, b.diffusion_coeff)


def test_mg_keeps_the_first_diagonal_after_update_params():
    """Quirk Q3 inside MG: the smoother's diagonal is the one of the first call."""
    rng = np.random.default_rng(5)
    x0, rhs = rng.random((40, 40)), rng.random((40, 40))
    a = da.MG(depth=1, smoother_iterations=2, maxiter=1, mass_coeff=1.0, diffusion_coeff=0.5, dim=2)
    b = so.MG(depth=1, smoother_iterations=2, maxiter=1, mass_coeff=1.0, diffusion_coeff=0.5, dim=2)
    a(x0, rhs), b(x0, rhs)
    a.update_params(mass_coeff=2.0, diffusion_coeff=0.1)
    b.update_params(mass_coeff=2.0, diffusion_coeff=0.1)
    np.testing.assert_array_equal(a(x0, rhs), b(x0, rhs))


@pytest.mark.parametrize("name,kw", [
    ("cg_2d", dict(mass_coeff=1.0, diffusion_coeff=0.5, dim=2)),
    ("cg_het", dict(mass_coeff=0.7, diffusion_coeff="diff", dim=2)),
    ("cg_3d", dict(mass_coeff=1.0, diffusion_coeff=0.5, dim=3)),
])
def test_cg_golden_to_solver_tolerance(name, kw):
    kw = dict(kw)
    if kw["diffusion_coeff"] == "diff":
        kw["diffusion_coeff"] = g(name, "diff")
    out = da.CG(maxiter=500, tol=1e-10, **kw)(g(name, "x0"), g(name, "rhs"))
    np.testing.assert_allclose(out, g(name, "out"), rtol=0, atol=1e-8)
    res = g(name, "rhs") - so.CG(maxiter=1, tol=1e-10, **kw).apply(out)
    assert np.linalg.norm(res) <= 1.5e-10 * np.linalg.norm(g(name, "rhs"))


def test_cg_few_iterations_same_recurrence():
    out = da.CG(maxiter=3, tol=1e-12, mass_coeff=1.0, diffusion_coeff=0.5, dim=2)(g("cg_few", "x0"), g("cg_few", "rhs"))
    np.testing.assert_allclose(out, g("cg_few", "out"), rtol=1e-11, atol=1e-12)


def test_cg_float32_and_grid_spacing_vs_oracle():
    rng = np.random.default_rng(6)
    x0 = rng.random((96, 80)).astype(np.float32)
    rhs = rng.random((96, 80)).astype(np.float32)
    kw = dict(maxiter=300, tol=1e-5, mass_coeff=1.0, diffusion_coeff=0.8, dim=2)
    out = da.CG(**kw)(x0, rhs, h=0.5)
    ref = so.CG(**kw)(x0, rhs, h=0.5)
    assert out.dtype == np.float32
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-4)


@pytest.mark.parametrize("name,dim,kw,solver", [
    ("tvd_mg_f64", 2, dict(mu=0.1, max_num_iter=6), dict(depth=1, smoother_iterations=2, maxiter=1)),
    ("tvd_mg_f32", 2, dict(mu=0.1, max_num_iter=6), dict(depth=1, smoother_iterations=2, maxiter=1)),
    ("tvd_mg_3d", 3, dict(mu=0.05, omega=0.8, max_num_iter=4), dict(depth=1, smoother_iterations=2, maxiter=2)),
])
def test_tvd_with_mg_golden_bit_exact(name, dim, kw, solver):
    img = g(name, "img")
    out = da.split_bregman_tvd(img.copy(), dim=dim, solver=da.MG(**solver), **kw)
    assert out.dtype == img.dtype
    np.testing.assert_array_equal(out, g(name, "out"))


def test_tvd_with_cg_golden_to_tolerance():
    img = g("tvd_cg_f64", "img")
    out = da.split_bregman_tvd(img.copy(), mu=0.1, max_num_iter=5, solver=da.CG(maxiter=200, tol=1e-10))
    np.testing.assert_allclose(out, g("tvd_cg_f64", "out"), rtol=0, atol=1e-7)


def test_tvd_with_mg_isotropic_eps_vs_oracle():
    from darsia_amd.synthetic import synthetic_image
    img = np.ascontiguousarray(synthetic_image(180, 250, 3)[..., 1]).astype(np.float64)
    kw = dict(mu=0.15, omega=1.0, max_num_iter=30, eps=1e-3, isotropic=True)
    sv = dict(depth=2, smoother_iterations=2, maxiter=1)
    out = da.split_bregman_tvd(img.copy(), solver=da.MG(**sv), **kw)
    ref = tv.split_bregman_tvd(img.copy(), promotion="strict", solver=so.MG(**sv), **kw)
    np.testing.assert_allclose(out, ref, rtol=1e-13, atol=1e-14)  # isotropic: sqrt / division order, cf. G4


def test_tvd_with_mg_rejects_channel_images():
    from darsia_amd.synthetic import synthetic_image
    with pytest.raises(ValueError):
        da.split_bregman_tvd(synthetic_image(32, 32, 3), mu=0.1, max_num_iter=2, solver=da.MG(depth=1))


def test_h1_regularization_with_cg_and_mg():
    """H1_regularization(solver=CG / MG) as in examples/paper/pet_simulations_comparison_block_b.py:680-686."""
    rng = np.random.default_rng(9)
    vol = rng.random((24, 30, 20))
    out = da.H1_regularization(vol, mu=0.1, omega=0.7, dim=3, solver=da.CG(maxiter=10000, tol=1e-9))
    ref = so.CG(maxiter=10000, tol=1e-9, dim=3, mass_coeff=0.7, diffusion_coeff=0.1)(vol, 0.7 * vol)
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-8)
    img = rng.random((40, 56, 3))
    out = da.H1_regularization(img, mu=0.2, omega=1.0, dim=2, solver=da.MG(depth=1, smoother_iterations=3, maxiter=2))
    for c in range(3):
        ref = so.MG(depth=1, smoother_iterations=3, maxiter=2, dim=2, mass_coeff=1.0, diffusion_coeff=0.2)(
            np.ascontiguousarray(img[..., c]), 1.0 * np.ascontiguousarray(img[..., c]))
        np.testing.assert_array_equal(out[..., c], ref)


def test_device_resident_chaining_through_jacobi_mg_cg_and_h1():
    """The reference chains these in host memory (restoration/h1_regularization.py:39-73, utils/linear_solvers/*.py); an
    image that is in HBM already (DeviceArray) goes through them without crossing PCIe and comes back as a DeviceArray
    with the bits of the host-array call."""
    import darsia_amd as da

    rng = np.random.default_rng(3)
    for dtype in (np.float32, np.float64):
        x0 = rng.random((70, 90)).astype(dtype)
        rhs = rng.random((70, 90)).astype(dtype)
        d_x0, d_rhs = da.DeviceArray.from_host(x0), da.DeviceArray.from_host(rhs)
        for solver in (da.Jacobi(maxiter=5, mass_coeff=0.7, diffusion_coeff=0.3), da.Jacobi(maxiter=30, tol=1e-4, mass_coeff=0.7, diffusion_coeff=0.3),
                       da.MG(depth=2, smoother_iterations=3, maxiter=4, mass_coeff=0.7, diffusion_coeff=0.3),
                       da.CG(maxiter=50, tol=1e-6, mass_coeff=0.7, diffusion_coeff=0.3)):
            want = solver(x0, rhs)
            got = solver(d_x0, d_rhs)
            assert isinstance(got, da.DeviceArray) and got.dtype == want.dtype and np.array_equal(got.get(), want), type(solver).__name__
            assert np.array_equal(solver(d_x0, rhs).get(), want)       # a host right-hand side next to a device iterate
        # H1_regularization: gray and colour images, Jacobi / MG
        col = rng.random((40, 60, 3)).astype(dtype)
        for img, solver in ((x0, da.Jacobi(maxiter=7)), (col, da.Jacobi(maxiter=4)), (x0, da.MG(depth=2, smoother_iterations=2, maxiter=3)),
                            (x0, da.CG(maxiter=40, tol=1e-6))):
            want = da.H1_regularization(img, mu=0.4, omega=1.3, dim=2, solver=solver)
            got = da.H1_regularization(da.DeviceArray.from_host(img), mu=0.4, omega=1.3, dim=2, solver=solver)
            assert isinstance(got, da.DeviceArray) and np.array_equal(got.get(), want), type(solver).__name__
        out = da.H1_regularization(da.ScalarImage(da.DeviceArray.from_host(x0)), mu=0.4, omega=1.3, solver=da.Jacobi(maxiter=3))
        assert isinstance(out.img, da.DeviceArray)
        assert np.array_equal(out.img.get(), da.H1_regularization(x0, mu=0.4, omega=1.3, solver=da.Jacobi(maxiter=3)))
    with pytest.raises(NotImplementedError):
        da.H1_regularization(da.DeviceArray.from_host(col), mu=0.4, omega=1.3, solver=da.Jacobi(maxiter=9, tol=1e-3))


def test_jacobi_takes_array_likes_and_numpy_scalar_coefficients():
    """ADVICE (round 3): ``Jacobi.__call__`` coerces lists like the reference's NumPy expressions do; NumPy-scalar
    coefficients (strongly typed under NEP 50, unlike Python floats) stay with the native flow instead of being erased."""
    rng = np.random.default_rng(3)
    x0 = rng.random((12, 17)).astype(np.float32)
    rhs = (1.3 * x0).astype(np.float32)
    s = da.Jacobi(maxiter=3)
    s.update_params(mass_coeff=1.3, diffusion_coeff=0.4, dim=2)
    a = s(x0, rhs)
    b = s(x0.tolist(), rhs.tolist())           # lists: float64 arrays after np.asarray -> the float64 (native) solve
    assert a.dtype == np.float64 and b.dtype == np.float64
    assert np.array_equal(b, tv.jacobi(np.asarray(x0.tolist()), np.asarray(rhs.tolist()), 1.3, 0.4, dim=2, maxiter=3))
    assert np.array_equal(a, tv.jacobi(x0, rhs, 1.3, 0.4, dim=2, maxiter=3))          # the reference's NumPy-2 flow
    s32 = da.Jacobi(maxiter=3)
    s32.update_params(mass_coeff=np.float32(1.3), diffusion_coeff=np.float32(0.4), dim=2)
    c = s32(x0, rhs)
    assert c.dtype == np.float32                # float32 scalars keep a float32 array float32 in NumPy as well


@pytest.mark.parametrize("shape", [(61, 300), (40, 517, 3), (300, 96, 3), (30, 40, 1), (2, 2)])
@pytest.mark.parametrize("maxiter", [2, 3, 4, 5, 6, 7, 9, 20])
def test_float64_multi_sweep_passes_in_the_lds_row_pipeline(shape, maxiter, monkeypatch):
    """Jacobi(maxiter > 1) on float64 operands (2-D, 1 / 3 channels): up to five sweeps per pass in the sweeps pipeline's PLAIN
    form (dxh_np2_sweeps.h) -- against the oracle's sweeps (jacobi.py:107-114) and against k_jacobi_multi<double>
    (DXH_FUSED_TUNE bit 134217728); H1_regularization (h1_regularization.py:73-78) on the same path."""
    rng = np.random.default_rng(maxiter * 13 + shape[1])
    x0, rhs = rng.random(shape), rng.random(shape)
    want = tv.jacobi(x0, rhs, 0.7, 0.3, dim=2, maxiter=maxiter, promotion="strict")
    for tune in (None, "134217728"):
        monkeypatch.delenv("DXH_FUSED_TUNE", raising=False)
        if tune:
            monkeypatch.setenv("DXH_FUSED_TUNE", tune)
        got = da.Jacobi(maxiter=maxiter, mass_coeff=0.7, diffusion_coeff=0.3)(x0, rhs)
        assert got.dtype == want.dtype and np.array_equal(got, want), (tune, np.abs(got - want).max())
    monkeypatch.delenv("DXH_FUSED_TUNE", raising=False)
    if maxiter in (3, 9):
        out = da.H1_regularization(x0, mu=0.4, omega=1.3, solver=da.Jacobi(maxiter=maxiter))
        monkeypatch.setenv("DXH_FUSED_TUNE", "134217728")
        assert np.array_equal(out, da.H1_regularization(x0, mu=0.4, omega=1.3, solver=da.Jacobi(maxiter=maxiter)))
        monkeypatch.delenv("DXH_FUSED_TUNE", raising=False)
        # ... and inside the TV loop (more sweeps than the one-pass-per-iteration pipeline takes: maxiter = 9)
        ref = tv.split_bregman_tvd(x0, mu=0.1, ell=0.3, max_num_iter=3, jacobi_maxiter=maxiter)
        assert np.array_equal(da.split_bregman_tvd(x0, mu=0.1, ell=0.3, max_num_iter=3, solver=da.Jacobi(maxiter=maxiter)), ref)


@pytest.mark.parametrize("shape,dim", [((37, 70), 2), ((9, 21, 40), 3)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("which", ["omega", "mu", "both"])
def test_h1_regularization_with_array_valued_coefficients(shape, dim, dtype, which):
    """The reference's volume workload passes omega per voxel (examples/paper/pet_simulations_comparison_block_b.py:680-686);
    h1_regularization.py:48-76 hands the arrays to the solver as they are.  Until round 4 the Jacobi branch here took
    float(omega) / float(mu) (TypeError for arrays) -- found by tests/fuzz/combos_r4.py."""
    rng = np.random.default_rng(21)
    img = rng.random(shape).astype(dtype)
    om = (0.7 * (0.2 + 0.8 * rng.random(shape))).astype(dtype) if which in ("omega", "both") else 0.7
    mu = (0.3 * (0.2 + 0.8 * rng.random(shape))).astype(dtype) if which in ("mu", "both") else 0.3
    for m in (1, 4):
        want = tv.h1_regularization(img, mu, om, dim=dim, jacobi_maxiter=m, promotion="numpy2" if dtype == np.float32 else "strict")
        got = da.H1_regularization(img, mu=mu, omega=om, dim=dim, solver=da.Jacobi(maxiter=m))
        assert got.dtype == want.dtype and np.array_equal(got, want)
    with pytest.raises(NotImplementedError):
        da.H1_regularization(da.DeviceArray.from_host(img), mu=mu, omega=om, dim=dim)
