"""Developer driver for counter passes: the device patch estimator alone, 200 patches of two 4000 x 8000 uint8 frames."""
import sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.imageregistration import PhaseCorrelationEstimator

H, W = 4000, 8000
rng = np.random.default_rng(0)
a = (rng.random((H, W), dtype=np.float32) * 255).astype(np.uint8)
b = np.roll(a, (3, -4), axis=(0, 1))
patches = da.Patches(da.Image(a), [10, 20], rel_overlap=0.1)
rois = [patches.rois[i][j] for i in range(10) for j in range(20)]
d_a, d_b = da.DeviceArray.from_host(a), da.DeviceArray.from_host(b)
est = PhaseCorrelationEstimator()
for _ in range(3):
    out = est.find_effective_translations(d_a, d_b, rois)
print(sum(ok for _, ok in out), "of", len(out), "accepted; first shift", out[0][0][:, 2])
