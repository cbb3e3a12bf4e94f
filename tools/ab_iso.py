"""A/B of library builds on the isotropic RGB loop and Chambolle (interleaved child processes on one box)."""
import os, subprocess, sys
CHILD = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da, time
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(4000, 8000, 3)
st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(16, isotropic=True)
iso = min(st.run(100, isotropic=True).loop_ms / 100 for _ in range(4)); st.close()
d = da.DeviceArray.from_host(img)
da.denoise_tv_chambolle(d, weight=0.1, max_num_iter=10, eps=0.0)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); da.denoise_tv_chambolle(d, weight=0.1, max_num_iter=60, eps=0.0); ts.append((time.perf_counter() - t0) / 60 * 1e3)
print(f"iso {iso:.4f} ms/it   chambolle {min(ts):.4f} ms/it")
'''
for rep in range(3):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "default":
            env["DARSIA_HIP_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{lib:45s}", out.stdout.strip() or out.stderr[-400:], flush=True)
