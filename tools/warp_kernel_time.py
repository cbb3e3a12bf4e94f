"""Developer tool: run find_and_warp on a 4000 x 8000 gray frame with 10 x 20 patches a few times (for rocprofv3
--kernel-trace --stats)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
H, W = 4000, 8000
gray = np.random.default_rng(0).random((H, W)).astype(np.float32)
patches = da.Patches(da.Image(gray), [10, 20])
disp = lambda p: np.array([6 * np.sin(np.pi * p[:, 0] / W), 4 * np.sin(np.pi * p[:, 1] / H)])
for _ in range(5):
    da.PiecewisePerspectiveTransform().find_and_warp(patches, disp)
