"""Golden vectors of the reference's float32 dtype flow under NumPy < 2 (value-based casting).

TEST INFRASTRUCTURE.  Run in the build container only, with the interpreter that carries NumPy 1.x:

    /opt/conda/bin/python3.9 tests/golden/make_golden_np1.py        # NumPy 1.26.4 here

Why a second set: ``tvd_golden.npz`` was produced under NumPy 2.2, where ``float32_array / np.float64_scalar`` is
float64 (NEP 50) and the reference's Jacobi iterate of a float32 image therefore becomes float64
(``utils/linear_solvers/jacobi.py:101-114``).  Under NumPy 1.x the same expression stays float32 (value-based
casting), so the reference computes everything in the image dtype: that is the arithmetic of this repo's NATIVE
float32 kernels (``promotion="native"``, what ``bench.py``'s headline times) and of the oracle's
``promotion="strict"`` flow.  These vectors pin both to bits the reference itself produced.

The reference's modules are loaded exactly as in ``make_golden.py`` (same shim, same image generator, a fresh
``Jacobi`` per call); the cases are its float32 cases plus the regimes where the two flows part visibly (active
shrinkage: ``mu = 1``; ``ell = 1`` at ``mu = 0.1``).  The file holds inputs and expected outputs only.
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

from make_golden import _shim, noisy_blocks  # noqa: E402  (the shim that loads the reference file by file)


def main():
    if int(np.__version__.split(".")[0]) >= 2:
        raise SystemExit("run this generator under NumPy < 2 (here: /opt/conda/bin/python3.9)")
    da = _shim()
    out = {}
    meta = {"numpy": np.__version__, "python": sys.version.split()[0]}

    def run(name, img, fresh_solver=None, capture_iters=False, **kw):
        solver = fresh_solver if fresh_solver is not None else da.Jacobi()
        buf = io.StringIO()
        if capture_iters:
            kw["verbose"] = True
        with contextlib.redirect_stdout(buf):
            res = da.split_bregman_tvd(img.copy(), solver=solver, **kw)
        assert res.dtype == img.dtype
        out[name + "/img"] = img
        out[name + "/out"] = res
        if capture_iters:
            n = sum(1 for ln in buf.getvalue().splitlines() if ln.startswith("Split Bregman iteration"))
            out[name + "/iters"] = np.int64(n)
        return res

    img32 = noisy_blocks((64, 96), 1, np.float64).astype(np.float32)
    rgb32 = noisy_blocks((48, 64, 3), 2, np.float32)

    # the float32 cases of make_golden.py (g2 / g3 / g7 / g12), same inputs
    run("p2_aniso2d_f32", img32, mu=0.1, max_num_iter=20)
    run("p3_aniso_rgb_f32", rgb32, mu=0.1, max_num_iter=10)
    run("p3_aniso_rgb_odd_f32", noisy_blocks((37, 53, 3), 3, np.float32), mu=0.15, omega=0.7, max_num_iter=7)
    run("p3_aniso_2ch_odd_f32", noisy_blocks((21, 35, 2), 4, np.float32), mu=0.1, max_num_iter=6)
    run("p2_aniso2d_odd_f32", noisy_blocks((33, 67), 5, np.float32), mu=0.2, ell=0.3, max_num_iter=9)
    run("p2_tiny_f32", noisy_blocks((1, 9), 6, np.float32), mu=0.1, max_num_iter=4)
    run("p2_tiny_col_f32", noisy_blocks((9, 1), 6, np.float32), mu=0.1, max_num_iter=4)
    run("p7_eps_f32", img32, mu=0.1, max_num_iter=100, eps=1e-3, capture_iters=True)
    run("p7_eps_rgb_f32", rgb32, mu=0.1, max_num_iter=100, eps=2e-3, capture_iters=True)

    # regimes where the NumPy-2 flow and this one differ by 10 - 80 x rtol 1e-4 (active shrinkage)
    run("p20_mu1_rgb_f32", rgb32, mu=1.0, max_num_iter=60)
    run("p21_ell1_f32", img32, mu=0.1, ell=1.0, max_num_iter=100)
    run("p22_mu1_gray_f32", img32, mu=1.0, omega=0.5, max_num_iter=40)

    # float32 coefficient arrays (arrays are never value-cast: float32 throughout)
    mu_arr = (np.where(np.arange(96)[None, :] < 48, 0.05, 0.3) * np.ones((64, 1))).astype(np.float32)
    run("p5_het_mu_f32", img32, mu=mu_arr, omega=0.1, max_num_iter=12)
    out["p5_het_mu_f32/mu"] = mu_arr
    rng = np.random.default_rng(11)
    rng.random((64, 96))  # keep the stream of make_golden.py
    rgb_mu = (0.05 + 0.2 * rng.random((48, 64, 3))).astype(np.float32)
    run("p5_het_mu_rgb_f32", rgb32, mu=rgb_mu, max_num_iter=6)
    out["p5_het_mu_rgb_f32/mu"] = rgb_mu

    # several Jacobi sweeps, a volume
    run("p6_jacobi3_rgb_f32", rgb32, fresh_solver=da.Jacobi(maxiter=3), mu=0.1, max_num_iter=5)
    vol = np.random.default_rng(9).random((12, 14, 10)).astype(np.float32)
    run("p9_vol3d_f32", vol, mu=0.1, dim=3, max_num_iter=6)

    # warm start: 5 iterations, then 5 more from the returned state (interleaved d / b as the reference holds them)
    d = np.zeros((*img32.shape, 2), dtype=np.float32)
    b = np.zeros((*img32.shape, 2), dtype=np.float32)
    first = da.split_bregman_tvd(img32.copy(), mu=0.1, max_num_iter=5, x0=(img32.copy(), d, b), solver=da.Jacobi())
    out["p8_warm_f32/img"] = img32
    out["p8_warm_f32/first"] = first
    out["p8_warm_f32/d5"] = d.copy()
    out["p8_warm_f32/b5"] = b.copy()
    second = da.split_bregman_tvd(img32.copy(), mu=0.1, max_num_iter=5, x0=(first, d, b), solver=da.Jacobi())
    out["p8_warm_f32/second"] = second
    out["p8_warm_f32/d10"] = d.copy()
    out["p8_warm_f32/b10"] = b.copy()

    # config-1 plumbing case in float32: 512 x 512, 20 iterations (checksum + crops; the image is regenerated from its seed)
    big32 = noisy_blocks((512, 512), 12, np.float64).astype(np.float32)
    res32 = da.split_bregman_tvd(big32.copy(), mu=0.1, max_num_iter=20, solver=da.Jacobi())
    assert res32.dtype == np.float32
    out["p12_512/seed"] = np.int64(12)
    out["p12_512/sum_f32"] = np.float64(res32.astype(np.float64).sum())
    out["p12_512/crop_tl_f32"] = res32[:16, :16].copy()
    out["p12_512/crop_mid_f32"] = res32[248:264, 248:264].copy()
    out["p12_512/crop_br_f32"] = res32[-16:, -16:].copy()

    # Jacobi.__call__ and H1 regularisation on float32 operands: float32 out under NumPy 1.x
    x = da.Jacobi(maxiter=4)
    x.update_params(mass_coeff=1.3, diffusion_coeff=0.4, dim=2)
    sol = x(x0=img32.copy(), rhs=(np.float32(1.3) * img32))
    out["p10_jacobi4_f32/img"] = img32
    out["p10_jacobi4_f32/out"] = sol
    meta["p10_jacobi_dtype"] = str(sol.dtype)
    h1 = da.H1_regularization(rgb32.copy(), mu=0.4, omega=1.3, dim=2, solver=da.Jacobi(maxiter=4))
    out["p10_h1_rgb_f32/img"] = rgb32
    out["p10_h1_rgb_f32/out"] = h1

    for k, v in meta.items():
        out["meta/" + k] = np.array(v)
    path = os.path.join(HERE, "tvd_np1_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")
    for k, v in meta.items():
        print(" ", k, "=", v)


if __name__ == "__main__":
    main()
