"""Developer timing: stages of ConcentrationAnalysis.map (upload on the uploader thread vs run + download on the caller's
thread), alone and overlapped."""
import sys, time, threading
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from concurrent.futures import ThreadPoolExecutor

H, W = 4000, 8000
rng = np.random.default_rng(0)
rgb = rng.random((H, W, 3)).astype(np.float32)
base_img = da.OpticalImage(rgb)
probe_img = da.OpticalImage(np.clip(rgb + 0.1 * rng.random(rgb.shape).astype(np.float32), 0, 1))
tvd_op = da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=100, eps=None)
model = da.CombinedModel([da.LinearModel(scaling=2.0, offset=0.0), da.ClipModel(min_value=0.0, max_value=1.0)])
ca = da.ConcentrationAnalysis(base_img, da.MonochromaticReduction(color="red"), None, tvd_op, model, **{"diff option": "positive"})
plan = ca._fusable(probe_img.img)
dt = ca._signal_dtype(probe_img.img, plan[0])
pool = ThreadPoolExecutor(1)
key = (H, W, dt.str)
c0 = ca._ctx_for(0, key, 3)
c1 = pool.submit(lambda: ca._ctx_for(1, key, 3)).result()

keep = []
def timed(f, *a):
    t0 = time.perf_counter(); keep.append(f(*a)); return (time.perf_counter() - t0) * 1e3

for rep in range(3):
    up = timed(ca._upload_frame, c0, probe_img.img, plan)
    fin = timed(ca._finish_frame, c0, plan)
    print(f"alone: upload {up:.2f} ms   run+model+download {fin:.2f} ms")
pool.submit(ca._upload_frame, c1, probe_img.img, plan).result()
for rep in range(3):
    t0 = time.perf_counter()
    fut = pool.submit(timed, ca._upload_frame, c1, probe_img.img, plan)
    fin = timed(ca._finish_frame, c0, plan)
    up = fut.result()
    print(f"overlapped: upload {up:.2f} ms   run+model+download {fin:.2f} ms   both {(time.perf_counter()-t0)*1e3:.2f} ms")
# pieces of finish
import ctypes as C
from darsia_amd import _lib
for rep in range(2):
    r = timed(lambda: c0.run(100, 1, None, eps=None, isotropic=False))
    out = np.empty((H, W), np.float32)
    e = timed(lambda: np.empty((H, W), np.float32).fill(0))
    d = timed(lambda: _lib.check(c0.lib.dxh_tvd_download_model(c0._h, _lib.ptr(out), None, 0)))
    d2 = timed(lambda: _lib.check(c0.lib.dxh_tvd_download_model(c0._h, _lib.ptr(out), None, 0)))
    print(f"run {r:.2f}  fresh-array touch {e:.2f}  download(fresh) {d:.2f}  download(touched) {d2:.2f}")

def series(n):
    t0 = time.perf_counter()
    res = list(ca.map([probe_img] * n))          # results kept, as a caller collecting a series would
    dt = (time.perf_counter() - t0) * 1e3 / n
    t0 = time.perf_counter(); del res
    print("   (dropping the %d results: %.2f ms each)" % (n, (time.perf_counter() - t0) * 1e3 / n))
    return dt
series(4)
print("map: %.2f ms/frame (8 frames)  %.2f ms/frame (16 frames)" % (series(8), series(16)))
t0 = time.perf_counter()
res = [ca(probe_img) for _ in range(8)]
print("frame by frame: %.2f ms/frame" % ((time.perf_counter() - t0) * 1e3 / 8))

def streaming(n):
    """Each result dropped before the next one is fetched, as a caller reducing every frame to a number would."""
    t0 = time.perf_counter()
    acc = 0.0
    for r in ca.map([probe_img] * n):
        acc += float(r.img[17, 17])
        del r
    return (time.perf_counter() - t0) * 1e3 / n
streaming(4)
print("map, results dropped as they come: %.2f ms/frame (8 frames)  %.2f ms/frame (16 frames)" % (streaming(8), streaming(16)))
