"""Developer timing: heterogeneous coefficients (mu array; the 'heterogeneous bregman' use case) vs scalar."""
import sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
H, W, C = 4000, 8000, 3
img = synthetic_image(H, W, C)
mu = np.full(img.shape, 0.1, np.float32); mu[:, W // 2:] = 0.3
for name, m, om, ell in (("scalar", 0.1, 1.0, None), ("mu array, ell = 2 mu", mu, 1.0, None),
                         ("mu array, scalar ell (TVD operator)", mu, 1.0, 1.0),
                         ("mu+omega arrays", mu, np.full(img.shape, 1.0, np.float32), None)):
    st = da.DeviceTVD(img.shape, img.dtype, dim=2); st.set_coeffs(m, om, ell, img); st.upload(img)
    st.run(60)
    ts = [st.run(60).loop_ms / 60 for _ in range(3)]
    print(f"{name:38s} {min(ts):.4f} ms/iter  {H*W/min(ts)/1e3:.0f} MPix/s", flush=True)
    st.close()
