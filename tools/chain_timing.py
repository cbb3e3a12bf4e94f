"""Developer timing: stages of the device-resident config-3 chain."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
H, W = 4000, 8000
rgb = np.random.default_rng(0).random((H, W, 3)).astype(np.float32)
u8 = (rgb * 255.0 + 0.5).astype(np.uint8)
A = np.eye(3) + 0.05 * np.random.default_rng(7).normal(size=(3, 3)); b = 0.01 * np.random.default_rng(7).normal(size=3)
def run():
    ts = [time.perf_counter()]
    d = da.DeviceArray.from_host(u8); ts.append(time.perf_counter())
    c = da.apply_affine_balance(d, A, b); ts.append(time.perf_counter())
    o = da.split_bregman_tvd(c, mu=0.1, max_num_iter=100, isotropic=True); ts.append(time.perf_counter())
    h = o.get(); ts.append(time.perf_counter())
    d.close(); c.close(); o.close(); ts.append(time.perf_counter())
    return [(b - a) * 1e3 for a, b in zip(ts, ts[1:])], h
for _ in range(4):
    t, h = run()
    print("from_host %.2f  affine %.2f  tvd %.2f  get %.2f  close %.2f   total %.2f ms" % (*t, sum(t)), flush=True)
    del h
