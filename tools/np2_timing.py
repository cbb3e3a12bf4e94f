"""Developer timing: the numpy2-promotion sweep (k_fused_np2) next to the native float32 and float64 pipelines."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image

for shape in ((4000, 8000, 3), (4000, 8000)):
    img = synthetic_image(shape[0], shape[1], 3 if len(shape) == 3 else 1)
    mu64 = 0.1 * (0.3 + 0.7 * np.random.default_rng(1).random(img.shape))
    for label, kw, mu in (("native f32 scalar", dict(), 0.1), ("numpy2 scalar", dict(promotion="numpy2"), 0.1),
                          ("numpy2 f64 mu array", dict(promotion="numpy2"), mu64),
                          ("native f32 mu array (KQV)", dict(), mu64.astype(np.float32)), ("native f64", dict(), 0.1)):
        im = img.astype(np.float64) if label == "native f64" else img
        with da.DeviceTVD(im.shape, im.dtype, **kw) as st:
            st.set_coeffs(mu, 1.0, 0.2, im)
            st.upload(im)
            st.run(8)
            r = min((st.run(40) for _ in range(3)), key=lambda r: r.loop_ms)
            print(f"{shape} {label:28s} {r.loop_ms / 40:.4f} ms/iteration  ({r.launches} launches)", flush=True)
    # several Jacobi sweeps in the numpy2 flow (three kinds of launches) next to the native float32 multi-sweep passes
    for label, kw in (("native f32 Jacobi(maxiter=5)", dict()), ("numpy2 Jacobi(maxiter=5)", dict(promotion="numpy2"))):
        with da.DeviceTVD(img.shape, img.dtype, **kw) as st:
            st.set_coeffs(0.1, 1.0, 0.2, img)
            st.upload(img)
            st.run(2, jacobi_maxiter=5)
            r = min((st.run(10, jacobi_maxiter=5) for _ in range(3)), key=lambda r: r.loop_ms)
            print(f"{shape} {label:28s} {r.loop_ms / 10:.4f} ms/iteration  ({r.launches} launches)", flush=True)
