"""Developer sweep: DXH_WAVES_PER_CU for several configurations of the K-iteration pipeline."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
H, W = 4000, 8000
rgb = synthetic_image(H, W, 3); gray = np.ascontiguousarray(rgb[..., 0])
cfgs = [("rgb f32 iso", rgb, True), ("gray f32", gray, False), ("gray f64", gray.astype(np.float64), False), ("rgb f64", rgb.astype(np.float64), False)]
for name, img, iso in cfgs:
    res = []
    for wpc in (8, 12, 16, 24, 32):
        os.environ["DXH_WAVES_PER_CU"] = str(wpc)
        st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(8, isotropic=iso)
        t = min(st.run(48, isotropic=iso).loop_ms for _ in range(3)) / 48
        res.append(f"{wpc}:{t:.4f}")
        st.close()
    print(name, "  ".join(res), flush=True)
