"""Developer sweep: row-chunk count (DXH_NCHUNKS, one process per value) of the numpy2 K = 4 pipelines, gray and RGB."""
import os, subprocess, sys
code = r'''
import sys, os
sys.path.insert(0, ".")
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
for ch in (1, 3):
    img = synthetic_image(4000, 8000, ch)
    if ch == 1: img = img.reshape(4000, 8000).copy()
    with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
        st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(8)
        r = min((st.run(100) for _ in range(3)), key=lambda r: r.loop_ms)
        print(f"nchunks={os.environ.get('DXH_NCHUNKS', 'model'):>6s} channels={ch}: {r.loop_ms / 100:.4f} ms/iteration", flush=True)
'''
for n in ("", "8", "16", "24", "32", "40", "48", "64", "80", "96", "128"):
    env = dict(os.environ)
    if n:
        env["DXH_NCHUNKS"] = n
    subprocess.run([sys.executable, "-c", code], env=env, check=True)
