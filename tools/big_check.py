"""Developer check: 16000 x 8000 x 3 float32 (1.5 GB per plane): the 4-iteration pipeline against single-iteration
sweeps (bit-exact), and non-finite pixels do not hang anything."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(16000, 8000, 3)
outs = {}
for k in ("4", "1"):
    os.environ["DXH_FUSE_K"] = k
    st = da.DeviceTVD(img.shape, img.dtype, dim=2); st.set_coeffs(0.1, 1.0, None, img); st.upload(img)
    r = st.run(12)
    outs[k] = st.download(); st.close()
    print("K", k, "ms/iter %.3f" % (r.loop_ms / 12), flush=True)
print("16000x8000x3 K=4 vs K=1:", "bit-exact" if np.array_equal(outs["4"], outs["1"]) else "MISMATCH", flush=True)
os.environ["DXH_FUSE_K"] = "4"
bad = synthetic_image(300, 500, 3); bad[10, 20, 1] = np.nan; bad[100, 200, 0] = np.inf; bad[200, 300, 2] = -np.inf
for f in (lambda: da.split_bregman_tvd(bad, mu=0.1, max_num_iter=9), lambda: da.split_bregman_tvd(bad, mu=0.1, max_num_iter=30, eps=1e-3),
          lambda: da.denoise_tv_chambolle(bad, weight=0.1, max_num_iter=20), lambda: da.denoise_tv_bregman(bad, weight=1.0, max_num_iter=10)):
    o = f(); print("non-finite input ->", o.shape, "finite fraction %.3f" % np.isfinite(o).mean(), flush=True)
