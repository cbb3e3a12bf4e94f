"""Developer trace: which thread does what when in ConcentrationAnalysis.map (wall-clock intervals of the upload, start, fetch
and wrap stages of every frame); under rocprofv3 --kernel-trace --memory-copy-trace the device side of the same run."""
import sys, time, threading
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
H, W = 4000, 8000
rng = np.random.default_rng(0)
rgb = rng.random((H, W, 3)).astype(np.float32)
base_img = da.OpticalImage(rgb)
probe_img = da.OpticalImage(np.clip(rgb + 0.1 * rng.random(rgb.shape).astype(np.float32), 0, 1))
tvd_op = da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=100, eps=None)
model = da.CombinedModel([da.LinearModel(scaling=2.0, offset=0.0), da.ClipModel(min_value=0.0, max_value=1.0)])
ca = da.ConcentrationAnalysis(base_img, da.MonochromaticReduction(color="red"), None, tvd_op, model, **{"diff option": "positive"})
log = []
def wrap(name):
    f = getattr(ca, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); log.append((name, threading.current_thread().name[:10], t0, time.perf_counter())); return r
    setattr(ca, name, g)
for n in ("_upload_frame", "_start_frame", "_fetch_frame", "_wrap"): wrap(n)
for r in ca.map([probe_img] * 4): del r
log.clear()
T0 = time.perf_counter()
for r in ca.map([probe_img] * 10): del r
T1 = time.perf_counter()
print("total %.2f ms/frame" % ((T1 - T0) * 100))
for name, th, a, b in sorted(log, key=lambda x: x[2]):
    print(f"{(a-T0)*1e3:8.2f} .. {(b-T0)*1e3:8.2f}  ({(b-a)*1e3:6.2f})  {th:10s} {name}")
