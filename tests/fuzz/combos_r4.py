"""Developer sweep (not random): every combination of scalar / array-valued mu, omega, ell (ell also absent) with scalars whose
quotients round differently in float32 and float64 (0.1, 0.3, 0.7), float32 images (native flow with float32 arrays, default
flow with float32 and with float64 arrays) and float64 images, dim = 2 and 3, one and three sweeps -- against the oracle, bit for bit.
usage: python tests/fuzz/combos_r4.py"""
import itertools, sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from oracle import tvd_numpy as O

rng = np.random.default_rng(9)
bad = n = 0
for shape, dim in (((37, 70), 2), ((23, 41, 3), 2), ((9, 21, 40), 3)):
    base = rng.random(shape)
    arr = {k: 0.2 + 0.8 * rng.random(shape) for k in ("mu", "omega", "ell")}
    for img_dt, arr_dt, promo in ((np.float32, np.float32, "native"), (np.float32, np.float32, None), (np.float32, np.float64, None), (np.float64, np.float64, None)):
        img = base.astype(img_dt)
        for mk, ok, ek, S in itertools.product(("s", "a"), ("s", "a"), ("s", "a", "none"), (1, 3)):
            kw = dict(dim=dim, max_num_iter=4)
            kw["mu"] = 0.1 if mk == "s" else (0.1 * arr["mu"]).astype(arr_dt)
            kw["omega"] = 0.7 if ok == "s" else (0.7 * arr["omega"]).astype(arr_dt)
            if ek != "none":
                kw["ell"] = 0.3 if ek == "s" else (0.3 * arr["ell"]).astype(arr_dt)
            if img_dt == np.float64:
                oracle_promo = "strict"
            elif promo == "native":
                oracle_promo = "strict"
            else:
                oracle_promo = "numpy2"
            try:
                ref = O.split_bregman_tvd(img, jacobi_maxiter=S, promotion=oracle_promo, **kw)
            except Exception as e:   # combinations the oracle does not take
                continue
            pk = {} if promo is None else dict(promotion=promo)
            out = da.split_bregman_tvd(img, solver=da.Jacobi(maxiter=S), **pk, **kw)
            n += 1
            if out.dtype != ref.dtype or not np.array_equal(out, ref):
                bad += 1
                print("MISMATCH", shape, "dim", dim, np.dtype(img_dt).name, "arrays", np.dtype(arr_dt).name, promo, "mu", mk, "omega", ok, "ell", ek, "S", S,
                      out.dtype, ref.dtype, "n", int(np.count_nonzero(out != ref)), "max", float(np.abs(out.astype(np.float64) - ref).max()), flush=True)
print(f"{n} combinations, {bad} mismatches")

# ---- the Jacobi operator and H1_regularization: scalar / array mass and diffusion coefficients, both float types, dim 2 / 3 ----
bad2 = n2 = 0
for shape, dim in (((37, 70), 2), ((9, 21, 40), 3)):
    base, r0 = rng.random(shape), rng.random(shape)
    arrs = {k: 0.2 + 0.8 * rng.random(shape) for k in ("m", "d")}
    for dt in (np.float32, np.float64):
        x0, rhs = base.astype(dt), r0.astype(dt)
        for mk, dk, m in itertools.product(("s", "a"), ("s", "a"), (1, 3, 7)):
            mass = 0.7 if mk == "s" else (0.7 * arrs["m"]).astype(dt)
            diff = 0.3 if dk == "s" else (0.3 * arrs["d"]).astype(dt)
            for promo in (("strict", "numpy2") if dt == np.float32 else ("strict",)):
                want = O.jacobi(x0, rhs, mass, diff, dim=dim, maxiter=m, promotion=promo)
                if dt == np.float32 and promo == "strict":
                    continue   # the operator follows the NumPy of the process (the reference's own behaviour): numpy2 here
                got = da.Jacobi(maxiter=m, dim=dim, mass_coeff=mass, diffusion_coeff=diff)(x0, rhs)
                n2 += 1
                if got.dtype != want.dtype or not np.array_equal(got, want):
                    bad2 += 1
                    print("MISMATCH Jacobi", shape, dim, np.dtype(dt).name, "mass", mk, "diffusion", dk, "m", m, got.dtype, want.dtype,
                          int(np.count_nonzero(got != want)), flush=True)
            if mk == "s":   # H1_regularization takes a scalar or array mu (diffusion) and omega (mass)
                want = O.h1_regularization(x0, diff, mass, dim=dim, jacobi_maxiter=m, promotion="numpy2" if dt == np.float32 else "strict")
                got = da.H1_regularization(x0, mu=diff, omega=mass, dim=dim, solver=da.Jacobi(maxiter=m))
                n2 += 1
                if got.dtype != want.dtype or not np.array_equal(got, want):
                    bad2 += 1
                    print("MISMATCH H1", shape, dim, np.dtype(dt).name, "mu", dk, "m", m, got.dtype, want.dtype, int(np.count_nonzero(got != want)), flush=True)
print(f"{n2} solver combinations, {bad2} mismatches")
sys.exit(1 if (bad or bad2) else 0)
