"""Developer: the three-launch numpy2 flow on 4000 x 8000 x 3 (five sweeps) and on a 256^3 volume, for a kernel trace."""
import sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(4000, 8000, 3)
with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
    st.set_coeffs(0.1, 1.0, 0.2, img); st.upload(img); st.run(2, jacobi_maxiter=5)
    r = st.run(6, jacobi_maxiter=5); print("2-D five sweeps", r.loop_ms / 6, "ms/iteration")
vol = np.random.default_rng(0).random((256, 256, 256), dtype=np.float32)
with da.DeviceTVD(vol.shape, vol.dtype, dim=3, promotion="numpy2") as st:
    st.set_coeffs(0.1, 1.0, None, vol); st.upload(vol); st.run(2)
    r = st.run(10); print("256^3 one sweep", r.loop_ms / 10, "ms/iteration")
