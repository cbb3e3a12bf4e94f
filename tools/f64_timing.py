"""Developer timing: float64 images through the K-stage pipelines -- dxh_np2_pipe.h with F = double against the generic
Pipe<double> (DXH_FUSED_TUNE bit 4194304)."""
import os, subprocess, sys
CHILD = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
out = []
for C in (3, 1):
    img = synthetic_image(4000, 8000, C).astype(np.float64)
    st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(12)
    best = min(st.run(96).loop_ms / 96 for _ in range(4))
    out.append(f"C={C} float64 {best:.4f} ms/it")
    st.close()
print(" | ".join(out))
'''
for rep in range(2):
    for tune in ("0", "4194304"):
        o = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, DXH_FUSED_TUNE=tune), capture_output=True, text=True)
        print("LDS-row pipeline (F = double)" if tune == "0" else "generic Pipe<double>        ", "|", o.stdout.strip() or o.stderr[-600:], flush=True)
