"""Developer timing: Chambolle iterations at full size (wall time difference of two iteration counts)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(4000, 8000, 3)
da.denoise_tv_chambolle(img[:64, :64], weight=0.1, eps=0.0, max_num_iter=2)
ts = {}
for n in (5, 205, 5, 205, 5, 205):
    t = time.perf_counter()
    da.denoise_tv_chambolle(img, weight=0.1, eps=0.0, max_num_iter=n)
    ts.setdefault(n, []).append(time.perf_counter() - t)
print("per iteration %.3f ms (RGB as volume); 5 its %.1f ms, 205 its %.1f ms" % ((min(ts[205]) - min(ts[5])) / 200 * 1e3, min(ts[5]) * 1e3, min(ts[205]) * 1e3))
g = np.ascontiguousarray(img[..., 0])
for n in (5, 205, 5, 205):
    t = time.perf_counter()
    da.denoise_tv_chambolle(g, weight=0.1, eps=0.0, max_num_iter=n)
    ts.setdefault(("g", n), []).append(time.perf_counter() - t)
print("per iteration %.3f ms (gray 2-D)" % ((min(ts[("g", 205)]) - min(ts[("g", 5)])) / 200 * 1e3))
