"""Developer A/B of library builds on the numpy2 (product default) flow: ms per iteration of the K-stage pipelines for each
DARSIA_HIP_LIB given (each in its own child process, interleaved twice).  python tools/ab_np2_libs.py default other.so[,ENV=V]"""
import os, subprocess, sys
CHILD = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
out = []
for C in (3, 1):
    img = synthetic_image(4000, 8000, C)
    mu64 = 0.1 * (0.3 + 0.7 * np.random.default_rng(1).random(img.shape))
    for label, mu, ell in (("scalar", 0.1, None), ("mu64", mu64, 0.2)):
        st = da.DeviceTVD(img.shape, img.dtype, promotion="numpy2"); st.set_coeffs(mu, 1.0, ell, img); st.upload(img); st.run(12)
        best = min(st.run(96).loop_ms / 96 for _ in range(4))
        out.append(f"C={C} {label} {best:.4f}")
        st.close()
print(" | ".join(out))
'''
libs = sys.argv[1:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ)
        lib, *sets = lib.split(",")  # lib.so,NAME=VALUE,...
        for kv in sets:
            k, v = kv.split("=")
            env[k] = v
        if lib != "default":
            env["DARSIA_HIP_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(lib, sets, out.stdout.strip() or out.stderr[-600:], flush=True)
