"""Developer: 60 iterations of the array-mu / scalar-ell pipeline on 4000 x 8000 x 3 float32 (for rocprofv3)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
rgb = synthetic_image(4000, 8000, 3)
mu = np.full(rgb.shape, 0.1, np.float32); mu[:, 4000:] = 0.3
st = da.DeviceTVD(rgb.shape, rgb.dtype); st.set_coeffs(mu, 1.0, 0.2, rgb); st.upload(rgb)
st.run(30); r = st.run(60); print(r.loop_ms / 60, r.launches); st.close()
