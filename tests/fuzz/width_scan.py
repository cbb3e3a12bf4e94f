import os, sys
os.environ.setdefault("DXH_PROMOTION", "native")   # these sweeps pin the native float32 kernels (oracle flow "strict")
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from oracle import tvd_numpy as O
rng = np.random.default_rng(0)
for dt in (np.float64, np.float32):
    for n_it in (2, 3, 4):
        badw = []
        for w in list(range(1, 40)) + list(range(90, 140)) + [200, 224, 225, 256, 300, 329, 447, 448, 449, 512]:
            img = rng.random((20, w)).astype(dt)
            out = da.split_bregman_tvd(img, mu=0.1, max_num_iter=n_it)
            ref = O.split_bregman_tvd(img, mu=0.1, max_num_iter=n_it, promotion="strict")
            if not np.array_equal(out, ref): badw.append(w)
        print(np.dtype(dt).name, "iters", n_it, "bad widths:", badw, flush=True)
