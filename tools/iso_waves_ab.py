"""Developer A/B: isotropic pipelines at 3 waves per SIMD (shipped) vs a -DDXH_ISO_WAVES=2 build (tools/_ab/)."""
import os, subprocess, sys
code = r'''
import sys, os
sys.path.insert(0, ".")
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
for shape in ((4000, 8000, 3), (4000, 8000)):
    img = synthetic_image(shape[0], shape[1], 3 if len(shape) == 3 else 1)
    with da.DeviceTVD(img.shape, img.dtype) as st:
        st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(12, isotropic=True)
        r = min((st.run(120, isotropic=True) for _ in range(3)), key=lambda r: r.loop_ms)
        print(f"{os.environ.get('DARSIA_HIP_LIB', 'shipped')[-24:]:26s} {shape} iso: {r.loop_ms / 120 * 4:.4f} ms per K = 4 launch", flush=True)
'''
for rep in range(2):
    for lib in (None, "tools/_ab/libdarsia_hip_iso2w.so"):
        env = dict(os.environ)
        if lib:
            env["DARSIA_HIP_LIB"] = os.path.abspath(lib)
        subprocess.run([sys.executable, "-c", code], env=env, check=True)
