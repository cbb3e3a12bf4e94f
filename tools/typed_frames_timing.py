"""Developer timing: ConcentrationAnalysis on 4000 x 8000 x 3 frames in the type they are stored in -- float32 frames alone,
float32 frames under a calibrated (float64) cleaning filter, uint8 frames -- per call and as a series (.map).  "widened first" is
what the path cost before the frames went over as stored: the host widens the frame to the signal's float64 and uploads that."""
import sys, time, warnings
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da

warnings.simplefilter("ignore")
H, W = 4000, 8000
rng = np.random.default_rng(0)
rgb = rng.random((H, W, 3), dtype=np.float32)
base = da.OpticalImage(rgb)
extra = da.OpticalImage(rgb + np.float32(0.01))
probe32 = np.clip(rgb + np.float32(0.1) * rng.random(rgb.shape, dtype=np.float32), 0, 1)
probe8 = (probe32 * 255).astype(np.uint8)
tvd_op = da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=100, eps=None)
model = da.CombinedModel([da.LinearModel(scaling=2.0, offset=0.0), da.ClipModel(min_value=0.0, max_value=1.0)])


def timeit(label, f, n=4):
    best, keep = 1e9, []   # results stay alive while timing: dropping a 128 - 256 MB array costs the host milliseconds of munmap
    for _ in range(n):
        t0 = time.perf_counter(); keep.append(f()); best = min(best, (time.perf_counter() - t0) * 1e3)
    print(f"{label:72s} {best:8.1f} ms", flush=True)


def series(ca, frames):
    t0 = time.perf_counter(); outs = list(ca.map(frames)); return (time.perf_counter() - t0) * 1e3 / len(frames)


for name, bases in (("no cleaning filter (float32 signal)", base), ("calibrated cleaning filter (float64 signal)", [base, extra])):
    ca = da.ConcentrationAnalysis(bases, da.MonochromaticReduction(color="red"), None, tvd_op, model, **{"diff option": "positive"})
    print(name)
    f32, u8 = da.OpticalImage(probe32), da.OpticalImage(probe8)
    timeit("  float32 frame, per call", lambda: ca(f32))
    if isinstance(bases, list):
        timeit("  float32 frame widened first on the host (the path before), per call", lambda: ca(da.OpticalImage(probe32.astype(np.float64))))
    timeit("  uint8 frame, per call", lambda: ca(u8))
    timeit("  uint8 frame converted first on the host (the path before), per call", lambda: ca(u8.img_as(float)))
    dev = da.OpticalImage(da.DeviceArray.from_host(probe32))
    timeit("  float32 frame already in HBM (DeviceArray in, DeviceArray out), per call", lambda: ca(dev))
    ca(f32); print(f"  float32 frames, series of 8: {min(series(ca, [f32] * 8) for _ in range(3)):8.1f} ms per frame")
    ca(u8); print(f"  uint8 frames, series of 8:   {min(series(ca, [u8] * 8) for _ in range(3)):8.1f} ms per frame")
    ca.close()
