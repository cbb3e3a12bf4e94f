"""Developer A/B helper: device-loop times of a few configurations for the library DARSIA_HIP_LIB points to."""
import sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
H, W = 4000, 8000
rgb = synthetic_image(H, W, 3); gray = np.ascontiguousarray(rgb[..., 0])
res = []
for name, img, iso in (("rgb aniso", rgb, False), ("rgb iso", rgb, True), ("gray aniso", gray, False)):
    st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(8, isotropic=iso)
    t = min(st.run(48, isotropic=iso).loop_ms for _ in range(4)) / 48
    res.append(f"{name} {t:.4f}")
    st.close()
print("  ".join(res))
