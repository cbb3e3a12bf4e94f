"""Developer sweep of DXH_NCHUNKS (row chunks of the K-iteration launches) per image kind; each setting in a child."""
import os, subprocess, sys
CHILD = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
kind = sys.argv[1]
H, W = 4000, 8000
img = synthetic_image(H, W, 3 if "rgb" in kind else 1)
if "f64" in kind: img = img.astype(np.float64)
st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(8)
print(f"{min(st.run(100).loop_ms / 100 for _ in range(4)):.4f}")
'''
kinds = sys.argv[1].split(",")
for kind in kinds:
    for nc in ["model"] + sys.argv[2:]:
        env = dict(os.environ)
        if nc != "model": env["DXH_NCHUNKS"] = nc
        out = subprocess.run([sys.executable, "-c", CHILD, kind], env=env, capture_output=True, text=True)
        print(kind, nc, out.stdout.strip() or out.stderr[-200:], flush=True)
