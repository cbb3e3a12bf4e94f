"""Developer timing: the stopping criterion on single-channel images, both flows (loop time, wall time, launches)."""
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da, time
from darsia_amd.synthetic import synthetic_image
img = np.ascontiguousarray(synthetic_image(4000, 8000, 3)[..., 0])
for promo in ("numpy2", "native"):
    with da.DeviceTVD(img.shape, img.dtype, promotion=promo) as st:
        st.set_coeffs(0.1, 1.0, 0.2, img); st.upload(img); st.run(12, eps=1e-30)
        for n in (96, 96):
            t0 = time.perf_counter(); r = st.run(n, eps=1e-30); wall = (time.perf_counter() - t0) * 1e3
            print(f"{promo} gray eps: {r.iters_done} its {r.loop_ms:.2f} ms loop ({wall:.2f} wall) = {r.loop_ms / r.iters_done:.4f} ms/it, {r.launches} launches")
        t0 = time.perf_counter(); r = st.run(96); wall = (time.perf_counter() - t0) * 1e3
        print(f"{promo} gray no eps: {r.loop_ms:.2f} ms loop = {r.loop_ms / 96:.4f} ms/it, {r.launches} launches")
