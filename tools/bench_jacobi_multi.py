"""Developer timing: Jacobi(maxiter = m) inside the TV loop, several sweeps per pass vs sweep by sweep."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(4000, 8000, 3)
for tune in ("0", "64"):
    os.environ["DXH_FUSED_TUNE"] = tune
    for m in (2, 3, 5, 20):
        st = da.DeviceTVD(img.shape, img.dtype, dim=2); st.set_coeffs(0.1, 1.0, None, img); st.upload(img)
        st.run(2, jacobi_maxiter=m)
        r = min(st.run(10, jacobi_maxiter=m).loop_ms for _ in range(2)) / 10
        print(f"{'multi-sweep passes' if tune == '0' else 'sweep by sweep    '} Jacobi(maxiter={m:2d}): {r:.3f} ms per outer iteration", flush=True)
        st.close()
