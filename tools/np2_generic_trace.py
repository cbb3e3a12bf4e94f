"""Developer run for rocprofv3 --kernel-trace --stats: the three-launch numpy2 flow, RGB with five Jacobi sweeps and a 256^3
volume with twenty (python3 tools/np2_generic_trace.py [rgb|vol])."""
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
which = sys.argv[1] if len(sys.argv) > 1 else "rgb"
if which == "rgb":
    img = synthetic_image(4000, 8000, 3)
    with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
        st.set_coeffs(0.1, 1.0, 0.2, img); st.upload(img)
        r = st.run(10, jacobi_maxiter=5)
else:
    vol = np.random.default_rng(0).random((256, 256, 256)).astype(np.float32)
    with da.DeviceTVD(vol.shape, vol.dtype, dim=3, promotion="numpy2") as st:
        st.set_coeffs(0.1, 1.0, None, vol); st.upload(vol)
        r = st.run(10, jacobi_maxiter=20)
print(which, r.loop_ms / 10, r.launches)
