"""Developer timing: float64 images (what scikit-image's img_as_float hands the reference's pipelines) through the TV loop in
the configurations users meet: scalar / array mu, isotropic, eps, several sweeps."""
import sys; sys.path.insert(0, ".")
import numpy as np, darsia_amd as da
from darsia_amd.synthetic import synthetic_image
rng = np.random.default_rng(0)
for C in (3, 1):
    img = synthetic_image(4000, 8000, 3).astype(np.float64)
    if C == 1: img = np.ascontiguousarray(img[..., 0])
    mu_arr = 0.1 * (0.3 + 0.7 * rng.random(img.shape))
    for name, mu, kw in (("scalar mu", 0.1, {}), ("array mu", mu_arr, {}), ("scalar mu, isotropic", 0.1, dict(isotropic=True)),
                         ("array mu, isotropic", mu_arr, dict(isotropic=True)), ("scalar mu, eps", 0.1, dict(eps=1e-9)),
                         ("scalar mu, Jacobi(3)", 0.1, dict(jacobi_maxiter=3)), ("array mu, omega, ell arrays", mu_arr, dict(arrs=True))):
        with da.DeviceTVD(img.shape, img.dtype) as st:
            if kw.get("arrs"): st.set_coeffs(mu, 0.8 + 0.4 * rng.random(img.shape), 0.2 + 0.2 * rng.random(img.shape), img)
            else: st.set_coeffs(mu, 1.0, 0.2, img)
            st.upload(img)
            rk = {k: v for k, v in kw.items() if k != "arrs"}
            st.run(8, **rk)
            r = min((st.run(48, **rk) for _ in range(3)), key=lambda r: r.loop_ms)
            print(f"float64 C={C} {name:32s} {r.loop_ms / max(1, r.iters_done):.3f} ms/it ({r.launches} launches, {r.iters_done} its)", flush=True)
