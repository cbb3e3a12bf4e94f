"""Developer timing: the three-launch numpy2 flow (several Jacobi sweeps; volumes) with and without the fused
shrinkage + next right-hand side + first sweep pass (DXH_FUSED_TUNE bit 524288 = the three launches)."""
import os, subprocess, sys
CHILD = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
out = []
img = synthetic_image(4000, 8000, 3)
for m in (5, 2):
    with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
        st.set_coeffs(0.1, 1.0, 0.2, img); st.upload(img); st.run(3, jacobi_maxiter=m)
        r = min((st.run(10, jacobi_maxiter=m) for _ in range(3)), key=lambda r: r.loop_ms)
        out.append(f"RGB Jacobi({m}) {r.loop_ms / 10:.3f} ms/it ({r.launches} launches)")
vol = np.random.default_rng(0).random((256, 256, 256)).astype(np.float32)
for m in (1, 20):
    with da.DeviceTVD(vol.shape, vol.dtype, dim=3, promotion="numpy2") as st:
        st.set_coeffs(0.1, 1.0, None, vol); st.upload(vol); st.run(3, jacobi_maxiter=m)
        r = min((st.run(10, jacobi_maxiter=m) for _ in range(3)), key=lambda r: r.loop_ms)
        out.append(f"256^3 Jacobi({m}) {r.loop_ms / 10:.3f} ms/it ({r.launches} launches)")
print(" | ".join(out))
'''
for rep in range(2):
    for tune in ("0", "524288"):
        env = dict(os.environ, DXH_FUSED_TUNE=tune)
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print("fused pass" if tune == "0" else "three launches", "|", o.stdout.strip() or o.stderr[-800:], flush=True)
