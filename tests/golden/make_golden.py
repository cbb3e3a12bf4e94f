"""Generate golden vectors for the TVD hot path by importing the reference.

TEST INFRASTRUCTURE.  Run in the build container only (``/root/reference`` is
not present on the GPU box):

    python tests/golden/make_golden.py

The reference package cannot be imported as a whole here (``cv2`` is missing,
an ordinary ``ModuleNotFoundError``), so only the hot-path modules are loaded
file by file onto an empty ``darsia`` shim module:

    utils/array_slice.py, utils/dtype.py, utils/derivatives.py,
    utils/linear_solvers/solver.py, utils/linear_solvers/jacobi.py,
    restoration/split_bregman_tvd.py, restoration/h1_regularization.py,
    restoration/tvd.py

``numba.njit`` is stubbed as the identity decorator (the isotropic shrinkage
then runs as the pure-Python loop it decorates) and ``skimage.img_as_float*``
as dtype pass-through/casts for float input (what skimage does for floats).

Every case uses a FRESH ``Jacobi`` instance: the reference caches its diagonal
on the solver object, so re-using the def-time default
instance would leak the first call's coefficients into later ones.

The output files hold inputs and expected outputs only (data, no source).
NumPy version used is recorded in the file; under NumPy >= 2 (NEP 50) the
reference's Jacobi step promotes float32 iterates to float64 because its
cached ``const_diag_scaled`` is a strongly typed ``np.float64`` scalar.
"""
from __future__ import annotations

import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/src/darsia"
HERE = os.path.dirname(os.path.abspath(__file__))


def _shim():
    da = types.ModuleType("darsia")
    sys.modules["darsia"] = da

    nb = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.njit = njit
    sys.modules["numba"] = nb

    sk = types.ModuleType("skimage")

    def _as_float(x):
        x = np.asarray(x)
        assert x.dtype.kind == "f", "fixtures are float-only"
        return x

    sk.img_as_float = _as_float
    sk.img_as_float32 = lambda x: np.asarray(x).astype(np.float32, copy=False)
    sk.img_as_float64 = lambda x: np.asarray(x).astype(np.float64, copy=False)
    rest = types.ModuleType("skimage.restoration")

    def _not_here(*a, **k):  # third-party methods are not part of these goldens
        raise NotImplementedError

    rest.denoise_tv_chambolle = _not_here
    rest.denoise_tv_bregman = _not_here
    sk.restoration = rest
    sys.modules["skimage"] = sk
    sys.modules["skimage.restoration"] = rest

    class Image:  # minimal stand-in so that tvd.py's isinstance checks work
        def __init__(self, img):
            self.img = img

        def copy(self):
            return Image(self.img.copy())

    da.Image = Image

    def load(rel):
        name = "darsia_ref_" + rel.replace("/", "_")[:-3]
        spec = importlib.util.spec_from_file_location(name, f"{REF}/{rel}")
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        for k, v in vars(m).items():
            if not k.startswith("_"):
                setattr(da, k, v)
        return m

    for rel in [
        "utils/array_slice.py",
        "utils/dtype.py",
        "utils/derivatives.py",
        "utils/linear_solvers/solver.py",
        "utils/linear_solvers/jacobi.py",
        "restoration/split_bregman_tvd.py",
        "restoration/h1_regularization.py",
        "restoration/tvd.py",
    ]:
        load(rel)
    return da


def noisy_blocks(shape, seed, dtype):
    """Piecewise-constant blocks + noise, clipped to [0, 1]."""
    rng = np.random.default_rng(seed)
    h, w = shape[:2]
    y, x = np.mgrid[0:h, 0:w]
    base = 0.5 + 0.25 * np.sign(np.sin(2 * np.pi * x / 32.0 + 0.3)) * np.sign(
        np.sin(2 * np.pi * y / 24.0 + 0.2)
    )
    if len(shape) == 3:
        gains = np.linspace(1.0, 0.6, shape[2])
        base = base[..., None] * gains
    img = base + rng.normal(0.0, 0.05, size=shape)
    return np.clip(img, 0.0, 1.0).astype(dtype)


def main():
    da = _shim()
    out = {}
    meta = {"numpy": np.__version__}

    def run(name, img, fresh_solver=None, capture_iters=False, **kw):
        solver = fresh_solver if fresh_solver is not None else da.Jacobi()
        buf = io.StringIO()
        if capture_iters:
            kw["verbose"] = True
        with contextlib.redirect_stdout(buf):
            res = da.split_bregman_tvd(img.copy(), solver=solver, **kw)
        out[name + "/img"] = img
        out[name + "/out"] = res
        if capture_iters:
            n = sum(1 for ln in buf.getvalue().splitlines() if ln.startswith("Split Bregman iteration"))
            out[name + "/iters"] = np.int64(n)
        return res

    # G1/G2 2-D anisotropic, scalar coefficients
    img64 = noisy_blocks((64, 96), 1, np.float64)
    img32 = img64.astype(np.float32)
    run("g1_aniso2d_f64", img64, mu=0.1, max_num_iter=20)
    run("g2_aniso2d_f32", img32, mu=0.1, max_num_iter=20)

    # G3 (H, W, 3) with dim=2: channel-coupled Jacobi (quirk Q2)
    rgb32 = noisy_blocks((48, 64, 3), 2, np.float32)
    run("g3_aniso_rgb_f32", rgb32, mu=0.1, max_num_iter=10)
    run("g3_aniso_rgb_f64", rgb32.astype(np.float64), mu=0.1, max_num_iter=10)
    # odd sizes (ragged vs. any vector width) and a 2-channel image
    run("g3_aniso_rgb_odd_f32", noisy_blocks((37, 53, 3), 3, np.float32), mu=0.15, omega=0.7, max_num_iter=7)
    run("g3_aniso_2ch_odd_f32", noisy_blocks((21, 35, 2), 4, np.float32), mu=0.1, max_num_iter=6)
    run("g2_aniso2d_odd_f32", noisy_blocks((33, 67), 5, np.float32), mu=0.2, ell=0.3, max_num_iter=9)
    run("g2_tiny_f32", noisy_blocks((1, 9), 6, np.float32), mu=0.1, max_num_iter=4)
    run("g2_tiny_col_f32", noisy_blocks((9, 1), 6, np.float32), mu=0.1, max_num_iter=4)
    run("g2_single_px_f32", noisy_blocks((1, 1), 6, np.float32), mu=0.1, max_num_iter=3)

    # G4 isotropic (reference: 2-D float64 only, numba signature)
    run("g4_iso2d_f64", img64[:40, :56].copy(), mu=0.1, max_num_iter=10, isotropic=True)

    # G5 heterogeneous coefficients
    mu_arr = np.where(np.arange(96)[None, :] < 48, 0.05, 0.3) * np.ones((64, 1))
    run("g5_het_mu_f64", img64, mu=mu_arr, omega=0.1, max_num_iter=12)
    out["g5_het_mu_f64/mu"] = mu_arr
    run("g5_het_mu_f32", img32, mu=mu_arr.astype(np.float32), omega=0.1, max_num_iter=12)
    out["g5_het_mu_f32/mu"] = mu_arr.astype(np.float32)
    rng = np.random.default_rng(11)
    om_arr = 0.5 + rng.random((64, 96))
    run("g5_het_omega_f64", img64, mu=0.1, omega=om_arr, ell=0.25, max_num_iter=8)
    out["g5_het_omega_f64/omega"] = om_arr
    rgb_mu = (0.05 + 0.2 * rng.random((48, 64, 3))).astype(np.float32)
    run("g5_het_mu_rgb_f32", rgb32, mu=rgb_mu, max_num_iter=6)
    out["g5_het_mu_rgb_f32/mu"] = rgb_mu

    # G6 inner solver variants
    run("g6_jacobi3_f64", img64, fresh_solver=da.Jacobi(maxiter=3), mu=0.1, max_num_iter=8)
    run("g6_jacobi3_rgb_f32", rgb32, fresh_solver=da.Jacobi(maxiter=3), mu=0.1, max_num_iter=5)
    run("g6_jacobi_tol_f64", img64, fresh_solver=da.Jacobi(maxiter=20, tol=1e-3), mu=0.1, max_num_iter=6)

    # G7 early stop
    run("g7_eps_f64", img64, mu=0.1, max_num_iter=100, eps=1e-3, capture_iters=True)
    run("g7_eps_f32", img32, mu=0.1, max_num_iter=100, eps=1e-3, capture_iters=True)
    run("g7_eps_rgb_f32", rgb32, mu=0.1, max_num_iter=100, eps=2e-3, capture_iters=True)

    # G14 adaptive regularisation (split_bregman_tvd.py:149-158); the Jacobi diagonal stays at its first
    # value (quirk Q3) while rhs and shrinkage follow the per-pixel ell
    run("g14_adaptive_f64", img64, mu=0.1, max_num_iter=9, adaptive=lambda it: it % 3 == 1)
    run("g14_adaptive_f32", img32, mu=0.1, max_num_iter=9, adaptive=lambda it: it % 3 == 1)
    run("g14_adaptive_rgb_f32", rgb32, mu=0.1, omega=0.9, max_num_iter=6, adaptive=lambda it: it >= 2)

    # G8 warm start: 5 iterations, then 5 more from the returned state
    for tag, im in (("f64", img64), ("f32", img32)):
        d = np.zeros((*im.shape, 2), dtype=im.dtype)
        b = np.zeros((*im.shape, 2), dtype=im.dtype)
        first = da.split_bregman_tvd(im.copy(), mu=0.1, max_num_iter=5, x0=(im.copy(), d, b), solver=da.Jacobi())
        out[f"g8_warm_{tag}/img"] = im
        out[f"g8_warm_{tag}/first"] = first
        out[f"g8_warm_{tag}/d5"] = d.copy()
        out[f"g8_warm_{tag}/b5"] = b.copy()
        second = da.split_bregman_tvd(im.copy(), mu=0.1, max_num_iter=5, x0=(first, d, b), solver=da.Jacobi())
        out[f"g8_warm_{tag}/second"] = second
        out[f"g8_warm_{tag}/d10"] = d.copy()
        out[f"g8_warm_{tag}/b10"] = b.copy()

    # G9 dim=3 volume
    vol = np.random.default_rng(9).random((12, 14, 10))
    run("g9_vol3d_f64", vol, mu=0.1, dim=3, max_num_iter=6)
    run("g9_vol3d_f32", vol.astype(np.float32), mu=0.1, dim=3, max_num_iter=6)

    # G10 H1 regularisation
    for tag, im in (("2d_f64", img64), ("2d_f32", img32), ("rgb_f32", rgb32)):
        res = da.H1_regularization(im.copy(), mu=0.4, omega=1.3, dim=2, solver=da.Jacobi(maxiter=4))
        out[f"g10_h1_{tag}/img"] = im
        out[f"g10_h1_{tag}/out"] = res

    # G11 derivatives
    a2 = np.random.default_rng(3).random((7, 9))
    a3 = np.random.default_rng(4).random((5, 6, 4)).astype(np.float32)
    out["g11_diff/a2"] = a2
    out["g11_diff/a3"] = a3
    for ax in range(2):
        out[f"g11_diff/a2_bwd{ax}"] = da.backward_diff(a2, ax)
        out[f"g11_diff/a2_fwd{ax}"] = da.forward_diff(a2, ax)
        out[f"g11_diff/a2_bwd{ax}_h"] = da.backward_diff(a2, ax, h=0.5)
        out[f"g11_diff/a3_bwd{ax}"] = da.backward_diff(a3, ax)
        out[f"g11_diff/a3_fwd{ax}"] = da.forward_diff(a3, ax)
    out["g11_diff/a3_bwd2"] = da.backward_diff(a3, 2, dim=3)
    out["g11_diff/a3_fwd2"] = da.forward_diff(a3, 2, dim=3)
    out["g11_diff/a2_lap"] = da.laplace(a2)
    out["g11_diff/a2_lap_h"] = da.laplace(a2, h=0.5, diffusion_coeff=2.0)
    out["g11_diff/a3_lap3"] = da.laplace(a3, dim=3)
    out["g11_diff/a2_lap_ax1"] = da.laplace(a2, axis=1)

    # G12 config-1 plumbing case: 512 x 512, 20 iterations (checksums + crops)
    big = noisy_blocks((512, 512), 12, np.float64)
    res = da.split_bregman_tvd(big.copy(), mu=0.1, max_num_iter=20, solver=da.Jacobi())
    out["g12_512/seed"] = np.int64(12)
    out["g12_512/sum"] = np.float64(res.sum())
    out["g12_512/sumsq"] = np.float64((res * res).sum())
    out["g12_512/crop_tl"] = res[:16, :16].copy()
    out["g12_512/crop_br"] = res[-16:, -16:].copy()
    out["g12_512/crop_mid"] = res[248:264, 248:264].copy()
    big32 = big.astype(np.float32)
    res32 = da.split_bregman_tvd(big32.copy(), mu=0.1, max_num_iter=20, solver=da.Jacobi())
    out["g12_512/sum_f32"] = np.float64(res32.astype(np.float64).sum())
    out["g12_512/crop_mid_f32"] = res32[248:264, 248:264].copy()

    # G13 TVD kwarg parsing (restoration/tvd.py:23-43)
    t = da.TVD(key="restoration ", **{
        "restoration method": "Heterogeneous Bregman", "restoration weight": 0.3,
        "restoration max_num_iter": 17, "restoration eps": 1e-5, "omega": 2.0, "dim": 2})
    meta["g13_attrs"] = repr((t.method, t.weight, t.max_num_iter, t.eps, t.omega, t.regularization, sorted(t.kwargs)))
    t2 = da.TVD()
    meta["g13_defaults"] = repr((t2.method, t2.weight, t2.max_num_iter, t2.eps, sorted(t2.kwargs)))
    # heterogeneous call through TVD: ell = regularization = 1.0 (not 2*mu)
    t3 = da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=6, eps=None, solver=da.Jacobi())
    out["g13_tvd_call/img"] = img32
    out["g13_tvd_call/out"] = t3(img32.copy())
    # quirk Q4: 'regularization' stays in kwargs and is forwarded twice
    t4 = da.TVD(method="heterogeneous bregman", regularization=0.5, solver=da.Jacobi())
    try:
        t4(img32.copy())
        meta["g13_q4"] = "no error"
    except TypeError as e:
        meta["g13_q4"] = "TypeError: " + str(e)
    try:
        da.TVD(method="nonsense")(img32)
        meta["g13_bad_method"] = "no error"
    except ValueError as e:
        meta["g13_bad_method"] = "ValueError: " + str(e)
    try:
        da.TVD()( [1, 2, 3] )
        meta["g13_bad_type"] = "no error"
    except TypeError as e:
        meta["g13_bad_type"] = "TypeError: " + str(e)

    for k, v in meta.items():
        out["meta/" + k] = np.array(v)
    path = os.path.join(HERE, "tvd_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")
    for k, v in meta.items():
        print(" ", k, "=", v)


if __name__ == "__main__":
    main()
