"""Developer check: device memory is returned after every kind of call (hipMemGetInfo before / after loops)."""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2**20
img = synthetic_image(600, 900, 3)
g = np.ascontiguousarray(img[..., 0])
mu = np.full(img.shape, 0.1, np.float32)
ops = {
    "tvd": lambda: da.split_bregman_tvd(img, mu=0.1, max_num_iter=8),
    "tvd eps het": lambda: da.split_bregman_tvd(img, mu=mu, max_num_iter=20, eps=1e-3),
    "tvd verbose-free adaptive": lambda: da.split_bregman_tvd(g, mu=0.1, max_num_iter=4, adaptive=lambda i: i % 2 == 0),
    "tvd MG": lambda: da.split_bregman_tvd(g, mu=0.1, max_num_iter=3, solver=da.MG(depth=2, smoother_iterations=2, maxiter=1)),
    "tvd CG": lambda: da.split_bregman_tvd(g, mu=0.1, max_num_iter=2, solver=da.CG(maxiter=20, tol=1e-6)),
    "chambolle": lambda: da.denoise_tv_chambolle(img, weight=0.1, max_num_iter=10),
    "bregman gs": lambda: da.denoise_tv_bregman(img, weight=1.0, max_num_iter=6, eps=1e-3),
    "jacobi/h1": lambda: da.H1_regularization(img, mu=0.3, omega=1.0, dim=2, solver=da.Jacobi(maxiter=7)),
    "mg": lambda: da.MG(depth=2, smoother_iterations=3, maxiter=2, mass_coeff=1.0, diffusion_coeff=0.4)(g, g),
    "resize": lambda: da.Resize(fx=0.5, fy=0.37, interpolation="inter_area")(img),
    "curvature": lambda: da.CurvatureCorrection(config={"bulge": dict(horizontal_bulge=1e-7)})(img),
    "affine": lambda: da.apply_affine_balance(img, np.eye(3), np.zeros(3)),
    "errors": lambda: [None for _ in range(1) if not _try()],
}
def _try():
    try:
        da.MG(depth=9, smoother_iterations=1, maxiter=1, mass_coeff=1.0, diffusion_coeff=1.0)(g[:5, :64].copy(), g[:5, :64].copy())
    except ValueError:
        return True
    return False
for name, f in ops.items():
    f(); b = free_mb()
    for _ in range(25): f()
    a = free_mb()
    print(f"{name:28s} free before {b:10.1f} MB after {a:10.1f} MB  delta {a - b:+.1f}", flush=True)
