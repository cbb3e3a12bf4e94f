"""Developer A/B of library builds on the two-waves-per-SIMD variants of dxh_np2_pipe.h: gray numpy2 (K = 6) and float64 images."""
import os, subprocess, sys
CHILD = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
out = []
img = synthetic_image(4000, 8000, 1)
st = da.DeviceTVD(img.shape, img.dtype, promotion="numpy2"); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(12)
out.append(f"gray numpy2 {min(st.run(96).loop_ms / 96 for _ in range(4)):.4f}"); st.close()
for C in (3, 1):
    img = synthetic_image(4000, 8000, C).astype(np.float64)
    st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(12)
    out.append(f"float64 C={C} {min(st.run(96).loop_ms / 96 for _ in range(4)):.4f}"); st.close()
print(" | ".join(out))
'''
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "default":
            env["DARSIA_HIP_LIB"] = os.path.abspath(lib)
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(lib, "|", o.stdout.strip() or o.stderr[-500:], flush=True)
