import os, sys
os.environ.setdefault("DXH_PROMOTION", "native")   # these sweeps pin the native float32 kernels (oracle flow "strict")
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from oracle import tvd_numpy as O
rng = np.random.default_rng(1)
# replay the generator to reach the failing cases
cases = []
for case in range(300):
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
    cases.append((img, kw, iso, jm))
shown = 0
for img, kw, iso, jm in cases:
    if iso or "eps" in kw: continue
    ref = O.split_bregman_tvd(img, isotropic=iso, jacobi_maxiter=jm, promotion="strict", **kw)
    out = da.split_bregman_tvd(img, isotropic=iso, solver=da.Jacobi(maxiter=jm), **kw)
    if np.array_equal(out, ref): continue
    n = kw["max_num_iter"]
    first = None
    for it in range(1, n + 1):
        k2 = dict(kw); k2["max_num_iter"] = it
        r = O.split_bregman_tvd(img, jacobi_maxiter=jm, promotion="strict", **k2)
        o = da.split_bregman_tvd(img, solver=da.Jacobi(maxiter=jm), **k2)
        if not np.array_equal(o, r):
            d = np.abs(o.astype(np.float64) - r.astype(np.float64))
            first = (it, float(d.max()), int(np.count_nonzero(d)), np.unravel_index(d.argmax(), d.shape), float(np.abs(r).max()))
            break
    print(img.shape, img.dtype, {k: (v if np.ndim(v) == 0 else "array") for k, v in kw.items()}, "jm", jm, "first diff at it", first, flush=True)
    shown += 1
    if shown >= 12: break
