import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
rgb = np.random.default_rng(0).random((4000, 8000, 3)).astype(np.float32)
op = da.TVD()
for _ in range(3):
    out = None
    t0 = time.perf_counter(); out = op(rgb); print("TVD() whole call %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
