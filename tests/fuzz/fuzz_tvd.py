"""Developer fuzz: random shapes / dtypes / parameters of split_bregman_tvd against the NumPy oracle, bit for bit
(anisotropic) or to 1e-6 / 1e-13 (isotropic: sqrt and division order, cf. golden G4)."""
import os, sys, time
os.environ.setdefault("DXH_PROMOTION", "native")   # these sweeps pin the native float32 kernels (oracle flow "strict")
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from oracle import tvd_numpy as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
t0 = time.time()
for case in range(n_cases):
    h, w = int(rng.integers(1, 90)), int(rng.integers(1, 700))
    c = int(rng.choice([0, 1, 2, 3, 4]))
    shape = (h, w) if c == 0 else (h, w, c)
    dt = rng.choice([np.float32, np.float64])
    img = rng.random(shape).astype(dt)
    kw = dict(max_num_iter=int(rng.integers(1, 14)))
    het = rng.random() < 0.35
    kw["mu"] = (0.02 + 0.2 * rng.random(shape)).astype(dt) if het else float(rng.choice([0.05, 0.1, 0.3]))
    if rng.random() < 0.3:
        kw["omega"] = (0.5 + rng.random(shape)).astype(dt) if (het and rng.random() < 0.5) else float(rng.choice([0.5, 1.0, 2.0]))
    if rng.random() < 0.3:
        kw["ell"] = float(rng.choice([0.3, 1.0]))
    iso = rng.random() < 0.3
    if rng.random() < 0.3:
        kw["eps"] = float(rng.choice([1e-2, 3e-3]))
    jm = int(rng.choice([1, 1, 1, 2, 3, 5]))
    ref = O.split_bregman_tvd(img, isotropic=iso, jacobi_maxiter=jm, promotion="strict", **kw)
    out = da.split_bregman_tvd(img, isotropic=iso, solver=da.Jacobi(maxiter=jm), **kw)
    if iso:
        ok = np.allclose(out, ref, rtol=1e-5 if dt == np.float32 else 1e-12, atol=1e-6 if dt == np.float32 else 1e-14)
    else:
        ok = np.array_equal(out, ref)
    if not ok:
        bad += 1
        print("MISMATCH", shape, np.dtype(dt).name, {k: (v if np.ndim(v) == 0 else "array") for k, v in kw.items()}, "iso", iso, "jm", jm,
              "max abs", float(np.abs(out.astype(np.float64) - ref.astype(np.float64)).max()), flush=True)
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
