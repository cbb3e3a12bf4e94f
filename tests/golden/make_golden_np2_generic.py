"""Golden vectors for float32 images OUTSIDE the default solve: several Jacobi sweeps per iteration, volumes (dim = 3),
more than four channels -- the configurations the device runs as three kinds of launches in the reference's own dtype flow
(promotion "numpy2": k_rhs_np2, k_jacobi_np2_first + the float64 sweeps, k_shrink_np2).

TEST INFRASTRUCTURE.  Run in the build container only (``/root/reference`` is not on the GPU box):

    python tests/golden/make_golden_np2_generic.py

What NumPy >= 2 does in the reference here (restoration/split_bregman_tvd.py:84-101,129-146; utils/linear_solvers/
jacobi.py:74-133): rhs and rhs / diag stay float32; the FIRST sweep of a call sums the neighbours of the float32 image in
float32 and divides by the np.float64 ``const_diag_scaled`` -> float64; every later sweep and iteration sums in float64;
d and b are float32 arrays.  The reference's modules are loaded exactly as in ``make_golden.py`` (same shim, a FRESH
``Jacobi`` per call).  The output holds inputs and expected outputs only.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _shim, noisy_blocks  # noqa: E402
from make_golden_mixed import weight_map  # noqa: E402


def main():
    da = _shim()
    out = {}
    meta = {"numpy": np.__version__}

    gray = noisy_blocks((64, 96), 41, np.float32)
    rgb = noisy_blocks((48, 64, 3), 42, np.float32)
    vol = noisy_blocks((20, 24, 28), 43, np.float32)
    six = noisy_blocks((40, 36, 6), 45, np.float32)           # 2-D image with six channels

    def run(name, img, maxiter, **kw):
        res = da.split_bregman_tvd(img.copy(), solver=da.Jacobi(maxiter=maxiter), **kw)
        out[name + "/img"] = img
        out[name + "/out"] = res
        out[name + "/maxiter"] = np.int64(maxiter)
        meta[name + "_dtype"] = str(res.dtype)
        return res

    # several sweeps, 2-D: 2 (first sweep float32 + one float64), 3, 5 (the 2-D multi-sweep passes), 6
    run("n1_gray_m2", gray, 2, mu=0.1, max_num_iter=12)
    run("n2_gray_m3_mu1", gray, 3, mu=1.0, omega=0.7, ell=0.45, max_num_iter=15)
    run("n3_rgb_m5", rgb, 5, mu=0.1, max_num_iter=10)
    run("n4_rgb_m6_ell", rgb, 6, mu=0.2, ell=0.3, max_num_iter=8)
    mu_g = 0.1 * weight_map(gray.shape, 46)
    out["n5_gray_m4_muarr/mu"] = mu_g
    run("n5_gray_m4_muarr", gray, 4, mu=mu_g, ell=0.5, max_num_iter=10)
    # volumes
    run("n6_vol_m1", vol, 1, mu=0.1, dim=3, max_num_iter=10)
    run("n7_vol_m4", vol, 4, mu=0.3, omega=0.5, ell=0.4, dim=3, max_num_iter=8)
    mu_v = 0.2 * (0.3 + 0.7 * np.random.default_rng(47).random(vol.shape))
    out["n8_vol_m2_muarr/mu"] = mu_v
    run("n8_vol_m2_muarr", vol, 2, mu=mu_v, ell=0.6, dim=3, max_num_iter=8)
    run("n9_vol_m7", vol, 7, mu=0.1, dim=3, max_num_iter=5)
    # six channels, one sweep (outside the fused pipelines' channel counts)
    run("n10_six_m1", six, 1, mu=0.1, max_num_iter=9)
    run("n11_six_m3", six, 3, mu=0.15, ell=0.5, max_num_iter=7)
    # early stop through eps with several sweeps
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = da.split_bregman_tvd(gray.copy(), solver=da.Jacobi(maxiter=3), mu=0.1, max_num_iter=200, eps=2e-3, verbose=True)
    out["n12_gray_m3_eps/img"] = gray
    out["n12_gray_m3_eps/out"] = res
    out["n12_gray_m3_eps/maxiter"] = np.int64(3)
    out["n12_gray_m3_eps/iters"] = np.int64(sum(1 for ln in buf.getvalue().splitlines() if ln.startswith("Split Bregman iteration")))

    # a tolerance in the inner solver (Q7: the iterate before the one that met it is returned)
    run_tol = lambda name, img, **kw: (out.__setitem__(name + "/img", img),                                     # noqa: E731
                                       out.__setitem__(name + "/out", da.split_bregman_tvd(img.copy(), **kw)))
    run_tol("t1_gray_tol", gray, solver=da.Jacobi(maxiter=20, tol=1e-3), mu=0.1, max_num_iter=6)
    run_tol("t2_rgb_tol", rgb, solver=da.Jacobi(maxiter=15, tol=3e-3), mu=0.2, ell=0.5, max_num_iter=5)
    run_tol("t3_vol_tol", vol, solver=da.Jacobi(maxiter=12, tol=2e-3), mu=0.1, dim=3, max_num_iter=4)
    run_tol("t4_gray_tol_loose", gray, solver=da.Jacobi(maxiter=5, tol=0.5), mu=0.1, max_num_iter=3)   # met at once: x0 comes back

    # Jacobi.__call__ on float32 operands (returns float64 under NumPy >= 2) and H1_regularization (float32 out)
    def jac(name, x0, rhs, solver, **params):
        solver.update_params(**params)
        res = solver(x0=x0.copy(), rhs=rhs.copy())
        out[name + "/x0"], out[name + "/rhs"], out[name + "/out"] = x0, rhs, res
        meta[name + "_dtype"] = str(res.dtype)

    rhs_g = noisy_blocks(gray.shape, 51, np.float32)
    jac("j1_gray_m1", gray, rhs_g, da.Jacobi(maxiter=1), mass_coeff=1.0, diffusion_coeff=0.2, dim=2)
    jac("j2_gray_m7", gray, rhs_g, da.Jacobi(maxiter=7), mass_coeff=0.7, diffusion_coeff=0.35, dim=2)
    jac("j3_rgb_m4", rgb, noisy_blocks(rgb.shape, 52, np.float32), da.Jacobi(maxiter=4), mass_coeff=1.0, diffusion_coeff=0.2, dim=2)
    jac("j4_vol_m3", vol, noisy_blocks(vol.shape, 53, np.float32), da.Jacobi(maxiter=3), mass_coeff=1.3, diffusion_coeff=0.4, dim=3)
    jac("j5_gray_tol", gray, rhs_g, da.Jacobi(maxiter=30, tol=1e-3), mass_coeff=1.0, diffusion_coeff=0.2, dim=2)

    def h1(name, img, solver, **kw):
        res = da.H1_regularization(img.copy(), solver=solver, **kw)
        out[name + "/img"], out[name + "/out"] = img, res
        meta[name + "_dtype"] = str(res.dtype)

    h1("h1_gray_m4", gray, da.Jacobi(maxiter=4), mu=0.4, omega=1.3, dim=2)
    h1("h2_rgb_m4", rgb, da.Jacobi(maxiter=4), mu=0.4, omega=1.3, dim=2)
    h1("h3_rgb_tol", rgb, da.Jacobi(maxiter=30, tol=1e-3), mu=1.0, omega=1.0, dim=2)
    h1("h4_vol_m6", vol, da.Jacobi(maxiter=6), mu=0.5, omega=0.8, dim=3)
    h1("h5_gray_m1", gray, da.Jacobi(), mu=1.0, omega=1.0, dim=2)

    # array-valued omega / ell (or ell = 2 mu) next to the float32 image.  a*: float64 arrays (parts of the right-hand side
    # turn float64); f*: float32 arrays only (every quantity stays float32)
    rng = np.random.default_rng(61)

    def arr(shape, lo, hi, dt=np.float64):
        return (lo + (hi - lo) * rng.random(shape)).astype(dt)

    def run_arr(name, img, maxiter=1, **kw):
        for k, v in kw.items():
            if isinstance(v, np.ndarray):
                out[f"{name}/{k}"] = v
        res = da.split_bregman_tvd(img.copy(), solver=da.Jacobi(maxiter=maxiter), **{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
        out[name + "/img"], out[name + "/out"], out[name + "/maxiter"] = img, res, np.int64(maxiter)
        meta[name + "_dtype"] = str(res.dtype)

    run_arr("a1_gray_omega", gray, mu=0.1, omega=arr(gray.shape, 0.5, 1.5), max_num_iter=10)
    run_arr("a2_gray_omega_ell", gray, mu=0.1, omega=arr(gray.shape, 0.5, 1.5), ell=0.4, max_num_iter=10)
    run_arr("a3_gray_ellarr", gray, mu=0.1, ell=arr(gray.shape, 0.15, 0.6), max_num_iter=10)
    run_arr("a4_rgb_all", rgb, mu=arr(rgb.shape, 0.02, 0.2), omega=arr(rgb.shape, 0.6, 1.2), ell=arr(rgb.shape, 0.2, 0.5), max_num_iter=8)
    run_arr("a5_vol_munone", vol, mu=arr(vol.shape, 0.03, 0.2), dim=3, max_num_iter=6)
    run_arr("a6_gray_omega_m3", gray, maxiter=3, mu=arr(gray.shape, 0.03, 0.2), omega=arr(gray.shape, 0.5, 1.5), ell=0.5, max_num_iter=7)
    run_arr("a7_vol_ellarr_m2", vol, maxiter=2, mu=0.2, ell=arr(vol.shape, 0.3, 0.8), omega=0.7, dim=3, max_num_iter=5)
    run_arr("f1_gray_munone_f32", gray, mu=arr(gray.shape, 0.03, 0.2, np.float32), max_num_iter=10)
    run_arr("f2_rgb_omega_f32", rgb, mu=0.1, omega=arr(rgb.shape, 0.5, 1.5, np.float32), max_num_iter=8)
    run_arr("f3_gray_all_f32_m2", gray, maxiter=2, mu=arr(gray.shape, 0.02, 0.2, np.float32), omega=arr(gray.shape, 0.6, 1.2, np.float32),
            ell=arr(gray.shape, 0.2, 0.5, np.float32), max_num_iter=6)

    # warm starts x0 = (u0, d0, b0): float32 d0 / b0 as a previous call left them (mutated in place by the reference), u0 the
    # previous float32 result -- or a float64 array (w3); recorded: the inputs of the SECOND call and what it returns / leaves
    def warm(name, img, first_iters, second_iters, maxiter=1, u0_as=None, **kw):
        dim = kw.get("dim", 2)
        d0 = np.zeros(img.shape + (dim,), np.float32)
        b0 = np.zeros(img.shape + (dim,), np.float32)
        cp = lambda k: {a: (v.copy() if isinstance(v, np.ndarray) else v) for a, v in k.items()}   # noqa: E731
        u1 = da.split_bregman_tvd(img.copy(), x0=(img.copy(), d0, b0), solver=da.Jacobi(maxiter=maxiter), max_num_iter=first_iters, **cp(kw))
        u0 = u1 if u0_as is None else (u1.astype(u0_as) * 1.0009765625).astype(u0_as)
        for k, v in kw.items():
            if isinstance(v, np.ndarray):
                out[f"{name}/{k}"] = v
        out[name + "/img"], out[name + "/u0"], out[name + "/d0"], out[name + "/b0"] = img, u0.copy(), d0.copy(), b0.copy()
        res = da.split_bregman_tvd(img.copy(), x0=(u0.copy(), d0, b0), solver=da.Jacobi(maxiter=maxiter), max_num_iter=second_iters, **cp(kw))
        out[name + "/out"], out[name + "/d"], out[name + "/b"] = res, d0.copy(), b0.copy()
        out[name + "/maxiter"], out[name + "/iters"] = np.int64(maxiter), np.int64(second_iters)
        meta[name + "_dtypes"] = f"{res.dtype} {d0.dtype} {u0.dtype}"

    warm("w1_gray", gray, 4, 5, mu=0.1)
    warm("w2_rgb_muarr_m2", rgb, 3, 4, maxiter=2, mu=arr(rgb.shape, 0.03, 0.2), ell=0.5)
    warm("w3_vol_u0_f64", vol, 3, 3, u0_as=np.float64, mu=0.2, omega=0.7, ell=0.4, dim=3)
    warm("w4_gray_one", gray, 2, 1, mu=1.0, ell=0.45)

    # adaptive regularisation (split_bregman_tvd.py:149-158) on float32 images: ell becomes a float32 array, the Jacobi
    # diagonals keep their cached values (quirk Q3).  Schedules are recorded as the list of iterations that adapt.
    def adapt(name, img, when, maxiter=1, **kw):
        for k, v in kw.items():
            if isinstance(v, np.ndarray):
                out[f"{name}/{k}"] = v
        res = da.split_bregman_tvd(img.copy(), solver=da.Jacobi(maxiter=maxiter), adaptive=lambda it: it in when,
                                   **{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
        out[name + "/img"], out[name + "/out"] = img, res
        out[name + "/when"], out[name + "/maxiter"] = np.array(sorted(when), dtype=np.int64), np.int64(maxiter)
        meta[name + "_dtype"] = str(res.dtype)

    adapt("d1_gray", gray, {1, 3}, mu=0.1, max_num_iter=6)
    adapt("d2_rgb_muarr", rgb, {0, 2, 3}, mu=arr(rgb.shape, 0.03, 0.2), omega=0.8, ell=0.6, max_num_iter=5)
    adapt("d3_vol_m2", vol, {1, 2}, maxiter=2, mu=0.2, omega=0.7, ell=0.4, dim=3, max_num_iter=4)
    adapt("d4_gray_late", gray, {4}, mu=1.0, ell=0.45, max_num_iter=6)

    # warm starts next to array-valued omega / ell (float64 arrays: parts of the right-hand side become float64, k_rhs_np2_het);
    # added in round 4, after everything else so that the shared generator of `arr` leaves the older vectors as they were
    warm("w5_gray_omega_ell_arr_m2", gray, 3, 4, maxiter=2, mu=0.1, omega=arr(gray.shape, 0.5, 1.5), ell=arr(gray.shape, 0.15, 0.6))
    warm("w6_vol_ellarr", vol, 2, 3, mu=0.2, omega=0.7, ell=arr(vol.shape, 0.3, 0.8), dim=3)
    warm("w7_rgb_munone_u0_f64", rgb, 2, 3, u0_as=np.float64, mu=arr(rgb.shape, 0.03, 0.2))

    for k, v in meta.items():
        out["meta/" + k] = np.array(v)
    path = os.path.join(HERE, "tvd_np2_generic_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")
    for k, v in meta.items():
        print(" ", k, "=", v)


if __name__ == "__main__":
    main()
