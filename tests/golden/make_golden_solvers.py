"""Golden vectors for the reference's MG and CG solvers (utils/linear_solvers/mg.py, cg.py) and for
``split_bregman_tvd(..., solver=MG(...)/CG(...))``.

TEST INFRASTRUCTURE.  The reference's MG / CG only run on NumPy < 2 and SciPy < 1.14 (``np.lib.pad``,
``cg(tol=)``); this image's default interpreter has NumPy 2.2 / SciPy 1.15, where both raise.  The conda
interpreter has NumPy 1.26.4 / SciPy 1.7.1:

    /opt/conda/bin/python3.9 tests/golden/make_golden_solvers.py

The reference modules are imported in place through the same shim as make_golden.py (nothing is copied);
only inputs and outputs are stored, labelled with the library versions.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as base  # noqa: E402


def main():
    warnings.simplefilter("ignore")
    import scipy

    da = base._shim()

    def load(rel):
        spec = importlib.util.spec_from_file_location("darsia_ref_" + rel.replace("/", "_")[:-3], f"{base.REF}/{rel}")
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        for k, v in vars(m).items():
            if not k.startswith("_"):
                setattr(da, k, v)

    load("utils/linear_solvers/mg.py")
    load("utils/linear_solvers/cg.py")
    out = {"meta/numpy": np.array(np.__version__), "meta/scipy": np.array(scipy.__version__)}
    rng = np.random.default_rng(42)

    def put(name, **arrs):
        for k, v in arrs.items():
            out[f"{name}/{k}"] = np.asarray(v)

    # ---- MG stand-alone ------------------------------------------------------------------------
    for tag, dt in (("f64", np.float64), ("f32", np.float32)):
        x0 = rng.random((33, 40)).astype(dt)
        rhs = rng.random((33, 40)).astype(dt)
        s = da.MG(depth=2, smoother_iterations=3, maxiter=2, mass_coeff=1.0, diffusion_coeff=0.5, dim=2)
        put(f"mg_2d_{tag}", x0=x0, rhs=rhs, out=s(x0.copy(), rhs.copy()))
    # depth 0, even sizes, one cycle
    x0 = rng.random((16, 24))
    rhs = rng.random((16, 24))
    put("mg_d0", x0=x0, rhs=rhs,
        out=da.MG(depth=0, smoother_iterations=2, maxiter=1, mass_coeff=0.3, diffusion_coeff=1.5, dim=2)(x0.copy(), rhs.copy()))
    # tolerance: returns the iterate before the converged one
    x0 = rng.random((21, 30))
    rhs = rng.random((21, 30))
    put("mg_tol", x0=x0, rhs=rhs,
        out=da.MG(depth=1, smoother_iterations=4, maxiter=50, tol=1e-3, mass_coeff=1.0, diffusion_coeff=0.2, dim=2)(x0.copy(), rhs.copy()))
    # 3-D volume
    x0 = rng.random((12, 15, 10))
    rhs = rng.random((12, 15, 10))
    put("mg_3d", x0=x0, rhs=rhs,
        out=da.MG(depth=1, smoother_iterations=2, maxiter=2, mass_coeff=1.0, diffusion_coeff=0.7, dim=3)(x0.copy(), rhs.copy()))
    # heterogeneous coefficients, two calls on the same object (the arrays are smeared by every cycle)
    for tag, dt in (("f64", np.float64), ("f32", np.float32)):
        x0 = rng.random((25, 36)).astype(dt)
        rhs = rng.random((25, 36)).astype(dt)
        mass = (0.5 + rng.random((25, 36))).astype(dt)
        diff = (0.2 + rng.random((25, 36))).astype(dt)
        s = da.MG(depth=1, smoother_iterations=2, maxiter=2, mass_coeff=mass.copy(), diffusion_coeff=diff.copy(), dim=2)
        o1 = s(x0.copy(), rhs.copy())
        o2 = s(o1.copy(), rhs.copy())
        put(f"mg_het_{tag}", x0=x0, rhs=rhs, mass=mass, diff=diff, out1=o1, out2=o2, mass_after=s.mass_coeff,
            diff_after=s.diffusion_coeff)
    # channel arrays are not supported by the reference's MG
    try:
        da.MG(depth=1, smoother_iterations=1, maxiter=1, mass_coeff=1.0, diffusion_coeff=1.0, dim=2)(
            rng.random((8, 8, 3)), rng.random((8, 8, 3)))
        put("mg_channels", raises=np.array("none"))
    except Exception as e:  # noqa: BLE001
        put("mg_channels", raises=np.array(type(e).__name__))

    # ---- CG stand-alone (tolerance parity only) ------------------------------------------------------
    x0 = rng.random((24, 31))
    rhs = rng.random((24, 31))
    put("cg_2d", x0=x0, rhs=rhs,
        out=da.CG(maxiter=500, tol=1e-10, mass_coeff=1.0, diffusion_coeff=0.5, dim=2)(x0.copy(), rhs.copy()))
    diff = 0.2 + rng.random((24, 31))
    put("cg_het", x0=x0, rhs=rhs, diff=diff,
        out=da.CG(maxiter=500, tol=1e-10, mass_coeff=0.7, diffusion_coeff=diff.copy(), dim=2)(x0.copy(), rhs.copy()))
    x0 = rng.random((10, 12, 9))
    rhs = rng.random((10, 12, 9))
    put("cg_3d", x0=x0, rhs=rhs,
        out=da.CG(maxiter=500, tol=1e-10, mass_coeff=1.0, diffusion_coeff=0.5, dim=3)(x0.copy(), rhs.copy()))
    x0 = rng.random((20, 20))
    rhs = rng.random((20, 20))
    put("cg_few", x0=x0, rhs=rhs,
        out=da.CG(maxiter=3, tol=1e-12, mass_coeff=1.0, diffusion_coeff=0.5, dim=2)(x0.copy(), rhs.copy()))

    # ---- split Bregman with the solvers ---------------------------------------------------------------
    img64 = base.noisy_blocks((40, 52), 5, np.float64)
    img32 = img64.astype(np.float32)
    put("tvd_mg_f64", img=img64,
        out=da.split_bregman_tvd(img64.copy(), mu=0.1, max_num_iter=6,
                                 solver=da.MG(depth=1, smoother_iterations=2, maxiter=1)))
    put("tvd_mg_f32", img=img32,
        out=da.split_bregman_tvd(img32.copy(), mu=0.1, max_num_iter=6,
                                 solver=da.MG(depth=1, smoother_iterations=2, maxiter=1)))
    vol = base.noisy_blocks((16, 20), 6, np.float64)[..., None] * np.linspace(1.0, 0.5, 12)
    vol = np.ascontiguousarray(vol + np.random.default_rng(7).normal(0, 0.02, vol.shape))
    put("tvd_mg_3d", img=vol,
        out=da.split_bregman_tvd(vol.copy(), mu=0.05, omega=0.8, dim=3, max_num_iter=4,
                                 solver=da.MG(depth=1, smoother_iterations=2, maxiter=2)))
    put("tvd_cg_f64", img=img64,
        out=da.split_bregman_tvd(img64.copy(), mu=0.1, max_num_iter=5, solver=da.CG(maxiter=200, tol=1e-10)))

    path = os.path.join(HERE, "solvers_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
