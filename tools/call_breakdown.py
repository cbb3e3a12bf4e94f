"""Developer timing: where the wall time of one split_bregman_tvd call goes (config 2)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(4000, 8000, 3)
da.split_bregman_tvd(img[:64, :64].copy(), mu=0.1, max_num_iter=2)
for rep in range(3):
    t = [time.perf_counter()]
    st = da.DeviceTVD(img.shape, img.dtype, dim=2); t.append(time.perf_counter())
    st.set_coeffs(0.1, 1.0, None, img); t.append(time.perf_counter())
    st.upload(img); t.append(time.perf_counter())
    r = st.run(100); t.append(time.perf_counter())
    out = st.download(); t.append(time.perf_counter())
    st.close(); t.append(time.perf_counter())
    names = ["create", "set_coeffs", "upload", "run(100)", "download", "close"]
    print("  ".join(f"{n} {1e3*(b-a):.2f}" for n, a, b in zip(names, t[:-1], t[1:])), f"| total {1e3*(t[-1]-t[0]):.2f} ms  loop {r.loop_ms:.2f}", flush=True)
t0 = time.perf_counter(); o = da.split_bregman_tvd(img, mu=0.1, max_num_iter=100); print("split_bregman_tvd", 1e3*(time.perf_counter()-t0))
