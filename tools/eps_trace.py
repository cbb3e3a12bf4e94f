"""Developer run for rocprofv3: the default flow with the stopping criterion on (eps), 4000 x 8000 (x 3) float32
(python3 tools/eps_trace.py [gray])."""
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(4000, 8000, 3)
if len(sys.argv) > 1 and sys.argv[1] == "gray":
    img = np.ascontiguousarray(img[..., 0])
with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
    st.set_coeffs(0.1, 1.0, 0.2, img); st.upload(img)
    r = st.run(30, eps=1e-12)
    print(r.iters_done, r.loop_ms / r.iters_done, r.launches)
    st.upload(img); r = st.run(30)
    print(r.iters_done, r.loop_ms / r.iters_done, r.launches)
