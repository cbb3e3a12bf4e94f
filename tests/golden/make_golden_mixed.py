"""Golden vectors for the MIXED-dtype heterogeneous flow: float32 image x float64 weight map.

TEST INFRASTRUCTURE.  Run in the build container only (``/root/reference`` is not on the GPU box):

    python tests/golden/make_golden_mixed.py

This is what the reference's workflow produces by default: ``build_restoration`` turns porosity weights and ignore
masks into a float64 array (presets/workflows/restoration.py:90,118-138: ``(~ignore_mask).astype(float)``,
``np.multiply(weight, active_mask)``), images come out of the colour correction as float32
(corrections/color/colorcorrection.py:390-394), and ``split_bregman_tvd`` then iterates ``u`` in float64 while ``d`` and
``b`` keep the image's float32 (restoration/split_bregman_tvd.py:99-101, 129-146; the Jacobi iterate is float64 under
NumPy >= 2 in any case: utils/linear_solvers/jacobi.py:101-114, ``const_diag_scaled`` is an ``np.float64``).

The reference's modules are loaded exactly as in ``make_golden.py`` (same shim, a FRESH ``Jacobi`` per call).  The
output holds inputs and expected outputs only.
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _shim, noisy_blocks  # noqa: E402


def weight_map(shape2, seed, floor=0.0):
    """A porosity-like float64 map in [floor, 1] with an ignored (zero) rectangle and a two-level facies split."""
    rng = np.random.default_rng(seed)
    h, w = shape2
    por = 0.35 + 0.65 * rng.random((h, w))
    por[:, : w // 3] *= 0.4
    if floor == 0.0:
        por[h // 4: h // 2, w // 5: w // 2] = 0.0      # ignore mask: (~mask).astype(float) == 0
    return np.maximum(por, floor)


def main():
    da = _shim()
    out = {}
    meta = {"numpy": np.__version__}

    gray = noisy_blocks((64, 96), 21, np.float32)
    rgb = noisy_blocks((48, 64, 3), 22, np.float32)
    big = noisy_blocks((160, 192), 23, np.float32)

    def run(name, img, **kw):
        res = da.split_bregman_tvd(img.copy(), solver=da.Jacobi(), **kw)
        out[name + "/img"] = img
        out[name + "/out"] = res
        meta[name + "_dtype"] = str(res.dtype)
        return res

    # scalar ell (what TVD passes: ell = regularization), float64 mu with zeros
    mu_g = 0.1 * weight_map(gray.shape, 31)
    out["m1_gray_ell1/mu"] = mu_g
    run("m1_gray_ell1", gray, mu=mu_g, ell=1.0, max_num_iter=20)
    out["m2_gray_ell02/mu"] = mu_g
    run("m2_gray_ell02", gray, mu=mu_g, ell=0.2, omega=0.8, max_num_iter=20)
    # ell = None -> ell = 2 mu is a float64 ARRAY (mu > 0 needed: ell divides)
    mu_p = 0.1 * weight_map(gray.shape, 32, floor=0.05)
    out["m3_gray_ellnone/mu"] = mu_p
    run("m3_gray_ellnone", gray, mu=mu_p, max_num_iter=20)
    # colour, channel-coupled (quirk Q2); ell = 0.2 keeps the coupled sweep contractive
    mu_c = np.repeat((0.1 * weight_map(rgb.shape[:2], 33))[..., None], 3, axis=2)
    out["m4_rgb_ell02/mu"] = mu_c
    run("m4_rgb_ell02", rgb, mu=mu_c, ell=0.2, max_num_iter=20)
    mu_cp = np.repeat((0.1 * weight_map(rgb.shape[:2], 34, floor=0.05))[..., None], 3, axis=2)
    out["m5_rgb_ellnone/mu"] = mu_cp
    run("m5_rgb_ellnone", rgb, mu=mu_cp, max_num_iter=20)
    # 100 iterations at TVD's default regularization (ell = 1.0): the regime where float32 arithmetic drifts furthest
    mu_b = 0.1 * weight_map(big.shape, 35)
    out["m6_gray_ell1_100/mu"] = mu_b
    run("m6_gray_ell1_100", big, mu=mu_b, ell=1.0, max_num_iter=100)
    # early stop through eps on the mixed flow (ell = 0.2: at ell = 1.0 the single-sweep iteration never settles where mu = 0)
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = da.split_bregman_tvd(gray.copy(), solver=da.Jacobi(), mu=mu_g, ell=0.2, omega=0.8, max_num_iter=200, eps=2e-3,
                                   verbose=True)
    out["m7_gray_eps/img"] = gray
    out["m7_gray_eps/mu"] = mu_g
    out["m7_gray_eps/out"] = res
    out["m7_gray_eps/iters"] = np.int64(sum(1 for ln in buf.getvalue().splitlines() if ln.startswith("Split Bregman iteration")))

    # the operator as build_restoration makes it (TVD: regularization = 1.0 unless configured; Q4: passing the kwarg
    # crashes the reference, so the default is what can be recorded) on an array and through its weight argument
    t = da.TVD(method="heterogeneous bregman", weight=mu_g, max_num_iter=20, eps=None, omega=1.0, solver=da.Jacobi())
    out["m8_tvd_operator/img"] = gray
    out["m8_tvd_operator/mu"] = mu_g
    out["m8_tvd_operator/out"] = t(gray.copy())

    # all-float32 inputs at ell = 1.0, 100 iterations: the scalar-coefficient flow in the same regime
    run("m9_gray_scalar_ell1_100", big, mu=0.05, ell=1.0, max_num_iter=100)

    for k, v in meta.items():
        out["meta/" + k] = np.array(v)
    path = os.path.join(HERE, "tvd_mixed_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")
    for k, v in meta.items():
        print(" ", k, "=", v)


if __name__ == "__main__":
    main()
