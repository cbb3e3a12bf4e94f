"""Developer A/B of library builds: ms per iteration of the headline loop for each DARSIA_HIP_LIB given (each in
its own child process, interleaved twice)."""
import os, subprocess, sys
CHILD = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(4000, 8000, 3)
st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(8)
best = min(st.run(100).loop_ms / 100 for _ in range(5))
print(f"{best:.4f}")
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
        print(lib, sets, out.stdout.strip() or out.stderr[-300:], flush=True)
