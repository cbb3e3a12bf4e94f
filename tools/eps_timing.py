"""Developer timing: the heterogeneous-bregman operator with its default stopping criterion (eps = 2e-4, at most 200
iterations) on 4000 x 8000 x 3 float32, host array in / out."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(4000, 8000, 3)
op = da.TVD(method="heterogeneous bregman", weight=0.1, regularization=0.2)
for _ in range(3):
    out = None
    t0 = time.perf_counter(); out = op(img); dt = (time.perf_counter() - t0) * 1e3
    print(f"TVD(heterogeneous bregman, eps=2e-4) whole call {dt:.1f} ms", flush=True)
with da.DeviceTVD(img.shape, img.dtype) as st:
    st.set_coeffs(0.1, 1.0, 0.2, img); st.upload(img)
    r = st.run(200, eps=2e-4)
    print(f"  device loop: {r.iters_done} iterations, converged={bool(r.converged)}, {r.loop_ms:.2f} ms = {r.loop_ms / max(1, r.iters_done):.4f} ms/iteration, {r.launches} launches")
    st.upload(img); r = st.run(r.iters_done)
    print(f"  same count without the criterion: {r.loop_ms:.2f} ms")
for C in (3, 1):
    im = img if C == 3 else np.ascontiguousarray(img[..., 0])
    with da.DeviceTVD(im.shape, im.dtype, promotion="numpy2") as st:
        st.set_coeffs(0.1, 1.0, 0.2, im); st.upload(im)
        r = st.run(200, eps=2e-4)
        print(f"  default flow (numpy2) C={C}: {r.iters_done} iterations, converged={bool(r.converged)}, {r.loop_ms:.2f} ms = {r.loop_ms / max(1, r.iters_done):.4f} ms/iteration, {r.launches} launches")
        st.upload(im); r = st.run(r.iters_done)
        print(f"  same count without the criterion: {r.loop_ms:.2f} ms = {r.loop_ms / max(1, r.iters_done):.4f} ms/iteration")
