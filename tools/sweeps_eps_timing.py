"""Developer timing: Jacobi(maxiter=3) with the stopping criterion, float32 (default flow) and float64 images: the sweeps pass that
reduces the increment itself against three launches + a reduction per iteration (DXH_FUSED_TUNE bit 33554432)."""
import sys; sys.path.insert(0, ".")
import os
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(4000, 8000, 3)
for dt, promo in ((np.float32, "numpy2"), (np.float64, None)):
    im = img.astype(dt)
    for tune in (0, 33554432):
        os.environ["DXH_FUSED_TUNE"] = str(tune)
        kw = {} if promo is None else dict(promotion=promo)
        with da.DeviceTVD(im.shape, im.dtype, **kw) as st:
            st.set_coeffs(0.1, 1.0, 0.2, im); st.upload(im); st.run(2, jacobi_maxiter=3, eps=1e-12)
            r = min((st.run(10, jacobi_maxiter=3, eps=1e-12) for _ in range(3)), key=lambda r: r.loop_ms)
            print(f"{np.dtype(dt).name} Jacobi(3) with eps, tune {tune}: {r.loop_ms / r.iters_done:.3f} ms/it ({r.launches} launches, rel {r.rel_increment if hasattr(r, 'rel_increment') else ''})", flush=True)
