"""Developer timing: whole call of split_bregman_tvd on 4000 x 8000 x 3 float32, 100 iterations, by band count."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.restoration.split_bregman_tvd import solve_banded

H, W = 4000, 8000
rgb = np.random.default_rng(0).random((H, W, 3)).astype(np.float32)
def t(f, n=3):
    f(); best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0); del r
    return best * 1e3
import os
for nb, edge in ((3, 1.0), (3, 0.6), (4, 1.0), (4, 0.6), (4, 0.45), (4, 0.3), (5, 0.45), (5, 0.3), (6, 0.3)):
    os.environ["DXH_BAND_EDGE"] = str(edge)
    print(f"nbands {nb} edge {edge}: {t(lambda: solve_banded(rgb, 0.1, 1.0, None, 100, nbands=nb)):.2f} ms", flush=True)
os.environ.pop("DXH_BAND_EDGE")
for nb in (1, 2, 0):
    used = solve_banded(rgb, 0.1, 1.0, None, 100, nbands=nb, return_bands=True)[1]
    print(f"nbands {nb} (used {used}): {t(lambda: solve_banded(rgb, 0.1, 1.0, None, 100, nbands=nb)):.2f} ms", flush=True)
print(f"split_bregman_tvd(): {t(lambda: da.split_bregman_tvd(rgb, mu=0.1, max_num_iter=100)):.2f} ms")
print(f"isotropic: {t(lambda: da.split_bregman_tvd(rgb, mu=0.1, max_num_iter=100, isotropic=True)):.2f} ms")
g = np.ascontiguousarray(rgb[..., 0])
print(f"gray f32: {t(lambda: da.split_bregman_tvd(g, mu=0.1, max_num_iter=100)):.2f} ms")
g64 = g.astype(np.float64)
print(f"gray f64: {t(lambda: da.split_bregman_tvd(g64, mu=0.1, max_num_iter=100)):.2f} ms")
print(f"20 iterations RGB: {t(lambda: da.split_bregman_tvd(rgb, mu=0.1, max_num_iter=20)):.2f} ms")
