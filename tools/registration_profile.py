"""Developer profile: where the host time of ImageRegistration.__call__ goes (configs[3] size, float32 host frames)."""
import cProfile, pstats, sys, time, io
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image

H, W = 4000, 8000
dst = synthetic_image(H, W, 1)
src = np.roll(dst, (3, -5), axis=(0, 1))
reg = da.ImageRegistration(da.Image(dst), N_patches=[10, 20], rel_overlap=0.1)
a = da.Image(src)
keep = [reg(a) for _ in range(3)]
ts = []
for _ in range(5):
    t0 = time.perf_counter(); keep.append(reg(a)); ts.append(time.perf_counter() - t0)
print("whole call ms:", [round(1e3 * t, 2) for t in ts])
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    keep.append(reg(a))
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
