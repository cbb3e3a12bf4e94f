"""Developer fuzz: larger shapes (many strips and row chunks) for the TV pipelines, Chambolle and Gauss-Seidel."""
import os, sys, time
os.environ.setdefault("DXH_PROMOTION", "native")   # these sweeps pin the native float32 kernels (oracle flow "strict")
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from oracle import bregman_gs_numpy as BG, chambolle_numpy as CH, tvd_numpy as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
t0 = time.time()
for case in range(n_cases):
    h, w = int(rng.integers(100, 700)), int(rng.integers(200, 3000)); c = int(rng.choice([0, 3]))
    shape = (h, w) if c == 0 else (h, w, c)
    dt = rng.choice([np.float32, np.float64])
    img = rng.random(shape).astype(dt)
    n = int(rng.integers(2, 14))
    het = rng.random() < 0.4
    mu = (0.02 + 0.2 * rng.random(shape)).astype(dt) if het else 0.1
    kw = dict(mu=mu, max_num_iter=n)
    if rng.random() < 0.4: kw["eps"] = 3e-3
    ok = np.array_equal(da.split_bregman_tvd(img, **kw), O.split_bregman_tvd(img, promotion="strict", **kw))
    o, k1 = da.denoise_tv_chambolle(img, weight=0.1, eps=1e-3, max_num_iter=n, return_iters=True)
    r, k2 = CH.denoise_tv_chambolle(img, weight=0.1, eps=1e-3, max_num_iter=n, return_iters=True)
    ok2 = k1 == k2 and np.array_equal(o, r)
    ok3 = True
    if case % 3 == 0:
        o, k1 = da.denoise_tv_bregman(img, weight=1.0, max_num_iter=min(n, 5), eps=1e-3, isotropic=False, return_iters=True)
        r, k2 = BG.denoise_tv_bregman(img, weight=1.0, max_num_iter=min(n, 5), eps=1e-3, isotropic=False, return_iterations=True)
        ok3 = k1 == k2 and np.array_equal(o, r)
    if not (ok and ok2 and ok3):
        bad += 1
        print("MISMATCH", shape, np.dtype(dt).name, n, het, kw.get("eps"), ok, ok2, ok3, flush=True)
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
