import os, subprocess, sys
CHILD = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
out = []
for C in (3, 1):
    img = synthetic_image(4000, 8000, 3).astype(np.float64)
    if C == 1: img = np.ascontiguousarray(img[..., 0])
    for eps in (None, 1e-12):
        st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(12, eps=eps)
        best = min(st.run(96, eps=eps).loop_ms / 96 for _ in range(3))
        out.append(f"C={C} eps={eps} {best:.4f}")
        st.close()
print(" | ".join(out))
'''
for rep in range(2):
    for lib in ("default", "tools/_ab/libdarsia_hip_f64w3.so"):
        env = dict(os.environ)
        if lib != "default": env["DARSIA_HIP_LIB"] = os.path.abspath(lib)
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(lib, "|", o.stdout.strip() or o.stderr[-600:], flush=True)
