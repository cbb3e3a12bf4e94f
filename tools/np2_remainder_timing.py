"""Developer timing: iteration counts that are not multiples of four in the default (numpy2) flow -- the remainders run as two- /
three-iteration passes (DXH_FUSE_K=... caps K; compare with the round-3 behaviour of one iteration per remainder launch)."""
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(4000, 8000, 3)
st = da.DeviceTVD(img.shape, img.dtype, promotion="numpy2"); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(8)
for n in (30, 31, 50, 10, 7):
    r = min((st.run(n) for _ in range(3)), key=lambda r: r.loop_ms)
    print(f"{n:3d} iterations: {r.loop_ms:7.3f} ms ({r.launches} launches, {r.loop_ms / n:.4f} ms per iteration)", flush=True)
