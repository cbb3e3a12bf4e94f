"""Developer timing: forced band counts for the numpy2 whole call (dxh_tvd_solve_banded_typed)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from darsia_amd.restoration.split_bregman_tvd import solve_banded_numpy2
from darsia_amd.synthetic import synthetic_image

for shape in ((4000, 8000, 3), (4000, 8000)):
    img = synthetic_image(shape[0], shape[1], 3 if len(shape) == 3 else 1)
    for n_iter in (100, 30):
        for nb in (0, 1, 2, 3, 4, 5, 6):
            solve_banded_numpy2(img, 0.1, 1.0, None, n_iter, nbands=nb)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); _, used = solve_banded_numpy2(img, 0.1, 1.0, None, n_iter, nbands=nb, return_bands=True); ts.append(time.perf_counter() - t0)
            print(f"{shape} {n_iter} iterations, nbands {nb} (used {used}): {1e3 * min(ts):7.1f} ms", flush=True)
