"""Developer timing: k_march3d_v over patch widths (DXH_MARCH_M2) for two volumes."""
import os, subprocess, sys
CHILD = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
rng = np.random.default_rng(0)
out = []
for shape in ((256, 256, 256), (200, 300, 180)):
    vol = rng.random(shape).astype(np.float32)
    st = da.DeviceTVD(vol.shape, vol.dtype, dim=3); st.set_coeffs(0.1, 1.0, None, vol); st.upload(vol); st.run(3)
    ms = min(st.run(20).loop_ms for _ in range(3)) / 20
    out.append(f"{shape}: {ms:.4f}")
    st.close()
print("; ".join(out))
'''
for m2 in ("auto", "32", "48", "64", "68", "96", "128", "132", "184", "256"):
    env = dict(os.environ)
    if m2 != "auto":
        env["DXH_MARCH_M2"] = m2
    env["DXH_FUSED_TUNE"] = "512"
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    choice = sorted(set(ln for ln in r.stderr.splitlines() if "march3d_v" in ln))
    print("M2", m2, r.stdout.strip(), "|", " ; ".join(c.replace("[dxh] march3d_v ", "") for c in choice), flush=True)
