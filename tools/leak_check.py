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
    # round 2 entry points
    "tvd banded": lambda: solve_banded(img, 0.1, 1.0, None, 12, nbands=3),
    "tvd banded het": lambda: solve_banded(img, mu, 1.0, 0.2, 9, nbands=2),
    "tvd kqv": lambda: da.split_bregman_tvd(img, mu=mu, ell=0.2, max_num_iter=9),
    "tvd device chain": lambda: _chain(),
    "concentration call": lambda: ca(frames[0]),
    "concentration map": lambda: list(ca.map(frames)),
    "concentration chambolle": lambda: ca_ch(frames[1]),
    "tps pair / displacement": lambda: reg.displacement(),
    "registration": lambda: reg(frames_gray[1]),
    # round 3 entry points (the default dtype flow of float32 images and what came with it)
    "tvd native f32": lambda: da.split_bregman_tvd(img, mu=0.1, max_num_iter=8, promotion="native"),
    "tvd numpy2 sweeps tol": lambda: da.split_bregman_tvd(img, mu=0.1, max_num_iter=4, solver=da.Jacobi(maxiter=5, tol=1e-3)),
    "tvd numpy2 volume": lambda: da.split_bregman_tvd(vol, mu=0.1, dim=3, max_num_iter=3, solver=da.Jacobi(maxiter=2)),
    "tvd numpy2 arrays": lambda: da.split_bregman_tvd(g, mu=mu64, omega=mu64 * 8, max_num_iter=4),
    "tvd numpy2 warm start": lambda: da.split_bregman_tvd(g, mu=0.1, max_num_iter=3, x0=(g, np.zeros(g.shape + (2,), np.float32), np.zeros(g.shape + (2,), np.float32))),
    "tvd numpy2 banded": lambda: solve_banded_numpy2(img, 0.1, 1.0, None, 9, nbands=3),
    "jacobi numpy2": lambda: jac(g, g),
    "warp patches": lambda: da.PiecewisePerspectiveTransform().find_and_warp(da.Patches(g, [3, 4]), lambda p: np.ones((2, len(p)))),
    "roi extraction": lambda: da.extract_quadrilateral_ROI(img, pts_src=[[5, 4], [4, 590], [880, 592], [890, 3]], shape=(500, 700)),
}
from darsia_amd.restoration.split_bregman_tvd import solve_banded, solve_banded_numpy2
vol = synthetic_image(40, 60, 3)[..., 0][:, :, None].repeat(32, axis=2).copy()
mu64 = np.full(g.shape, 0.1)
jac = da.Jacobi(maxiter=4)
jac.update_params(mass_coeff=1.0, diffusion_coeff=0.3, dim=2)
base = da.OpticalImage(img)
frames = [da.OpticalImage(np.clip(img + 0.05 * k, 0, 1).astype(np.float32)) for k in range(1, 4)]
model = da.CombinedModel([da.LinearModel(scaling=2.0, offset=0.0), da.ClipModel(min_value=0.0, max_value=1.0)])
ca = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="red"), None,
                              da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=10, eps=None), model, **{"diff option": "positive"})
ca_ch = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="gray"), None, da.TVD(weight=0.1, max_num_iter=8), model)
frames_gray = [da.ScalarImage(g), da.ScalarImage(np.roll(g, 2, axis=1).copy())]
reg = da.ImageRegistration(frames_gray[0], N_patches=[3, 4])
reg(frames_gray[1])
def _chain():
    d = da.DeviceArray.from_host(img)
    out = da.split_bregman_tvd(d, mu=0.1, max_num_iter=4)
    res = out.get()
    d.close(); out.close()
    return res
def _try():
    try:
        da.MG(depth=9, smoother_iterations=1, maxiter=1, mass_coeff=1.0, diffusion_coeff=1.0)(g[:5, :64].copy(), g[:5, :64].copy())
    except ValueError:
        return True
    return False
for name, f in ops.items():
    f(); f(); b = free_mb()
    for _ in range(25): f()
    a = free_mb()
    print(f"{name:28s} free before {b:10.1f} MB after {a:10.1f} MB  delta {a - b:+.1f}", flush=True)
print(f"trim returned {da.trim_device_cache() / 2**20:.1f} MB; free now {free_mb():.1f} MB")
