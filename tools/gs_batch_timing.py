"""Developer timing: scikit-image's Gauss-Seidel split Bregman (TVD(method="isotropic Bregman")) on a 4000 x 8000 image for several
numbers of pipelined sweeps per batch (DXH_GS_BATCH)."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
rng = np.random.default_rng(0)
img = rng.random((4000, 8000)).astype(np.float32)
for kw in (dict(weight=0.025, eps=1e-4, max_num_iter=100), dict(weight=0.1, eps=2e-4, max_num_iter=200)):
    for b in ("64", "100", "128", "200", "256"):
        os.environ["DXH_GS_BATCH"] = b
        op = da.TVD(method="isotropic Bregman", **kw)
        op(img)
        t0 = time.perf_counter(); out = op(img); dt = (time.perf_counter() - t0) * 1e3
        print(kw, "batch", b, f"{dt:.1f} ms", float(out[7, 7]), flush=True)
