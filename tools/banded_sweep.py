"""Developer: whole-call time by band count for several workloads (is the library's automatic choice near the best?)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from darsia_amd.restoration.split_bregman_tvd import solve_banded
rng = np.random.default_rng(0)
def t(f, n=3):
    f(); best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0); del r
    return best * 1e3
cases = [("rgb f32 100 it", (4000, 8000, 3), np.float32, 100, False), ("rgb f32 iso 100 it", (4000, 8000, 3), np.float32, 100, True),
         ("gray f32 100 it", (4000, 8000), np.float32, 100, False), ("gray f64 100 it", (4000, 8000), np.float64, 100, False),
         ("rgb f32 20 it", (4000, 8000, 3), np.float32, 20, False), ("rgb f32 300 it", (4000, 8000, 3), np.float32, 300, False),
         ("rgb f32 2000x4000 100 it", (2000, 4000, 3), np.float32, 100, False), ("rgb f64 100 it", (4000, 8000, 3), np.float64, 100, False)]
for name, shape, dt, n, iso in cases:
    img = rng.random(shape).astype(dt)
    auto = solve_banded(img, 0.1, 1.0, None, n, 1, iso, return_bands=True)[1]
    row = [f"{nb}:{t(lambda: solve_banded(img, 0.1, 1.0, None, n, 1, iso, nbands=nb)):.1f}" for nb in (1, 2, 3, 4, 5, 6)]
    print(f"{name:28s} auto={auto} -> {t(lambda: solve_banded(img, 0.1, 1.0, None, n, 1, iso)):.1f} ms | " + "  ".join(row), flush=True)
