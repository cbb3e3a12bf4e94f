"""Developer timing: whole split_bregman_tvd() calls on host arrays, default (numpy2) and native flow."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image

for shape in ((4000, 8000, 3), (4000, 8000)):
    img = synthetic_image(shape[0], shape[1], 3 if len(shape) == 3 else 1)
    mu64 = 0.1 * (0.3 + 0.7 * np.random.default_rng(1).random(img.shape))
    for label, kw in (("default (numpy2), scalar mu", dict(mu=0.1)), ("default (numpy2), float64 mu array, ell = 1", dict(mu=mu64, ell=1.0)),
                      ("native, scalar mu", dict(mu=0.1, promotion="native")), ("native, float32 mu array, ell = 1", dict(mu=mu64.astype(np.float32), ell=1.0, promotion="native"))):
        da.split_bregman_tvd(img, max_num_iter=100, **kw)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); out = da.split_bregman_tvd(img, max_num_iter=100, **kw); ts.append(time.perf_counter() - t0)
        print(f"{shape} {label:46s} {1e3 * min(ts):7.1f} ms per call (100 iterations)", flush=True)
    op = da.TVD(method="heterogeneous bregman", weight=mu64, max_num_iter=100, eps=None)
    op(img)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); op(img); ts.append(time.perf_counter() - t0)
    print(f"{shape} TVD operator object, float64 weight map (context kept)   {1e3 * min(ts):7.1f} ms per call", flush=True)
