"""Developer A/B of library builds on the isotropic pipelines: ms per K = 4 launch for each DARSIA_HIP_LIB given (each in
its own child process, interleaved twice).  usage: python tools/ab_iso_libs.py default path/to/other.so"""
import os, subprocess, sys
CHILD = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
out = []
for shape in ((4000, 8000, 3), (4000, 8000)):
    img = synthetic_image(shape[0], shape[1], 3 if len(shape) == 3 else 1)
    st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(8, isotropic=True)
    r = min((st.run(100, isotropic=True) for _ in range(5)), key=lambda r: r.loop_ms)
    out.append(f"{shape}: {1e3 * r.loop_ms / r.launches:.1f} us/launch ({r.loop_ms / 100:.4f} ms/it)")
    st.close()
print("; ".join(out))
'''
libs = sys.argv[1:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default":
            env["DARSIA_HIP_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(lib, out.stdout.strip() or out.stderr[-300:], flush=True)
