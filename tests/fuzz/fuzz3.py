"""Developer fuzz, part 3: piecewise perspective warp, CurvatureCorrection, heterogeneous MG, CG."""
import os, sys, time
os.environ.setdefault("DXH_PROMOTION", "native")   # these sweeps pin the native float32 kernels (oracle flow "strict")
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from oracle import curvature_numpy as CU, solvers_numpy as SO, warp_numpy as W
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
def chk(name, a, b, info, tol=None):
    global bad
    ok = (a.shape == b.shape) and (np.array_equal(a, b) if tol is None else np.allclose(a, b, rtol=0, atol=tol))
    if not ok:
        bad += 1
        print("MISMATCH", name, info, "max abs", float(np.abs(a.astype(float) - b.astype(float)).max()) if a.shape == b.shape else (a.shape, b.shape), flush=True)
for case in range(n_cases):
    h, w = int(rng.integers(24, 120)), int(rng.integers(24, 200)); c = int(rng.choice([0, 3]))
    shape = (h, w) if c == 0 else (h, w, c)
    dtype = rng.choice([np.float32, np.float64, np.uint8])
    img = rng.random(shape)
    img = (img * 255).astype(np.uint8) if dtype == np.uint8 else img.astype(dtype)
    npatch = [int(rng.integers(1, 6)), int(rng.integers(1, 7))]
    overlap = float(rng.choice([0.0, 0.05, 0.2]))
    ax, ay = float(rng.uniform(-6, 6)), float(rng.uniform(-6, 6))
    disp = lambda p, ax=ax, ay=ay: np.array([ax * np.sin(np.pi * p[:, 1] / h) * np.sin(np.pi * p[:, 0] / w), ay * np.sin(np.pi * p[:, 0] / w)])
    rev = bool(rng.random() < 0.5)
    patches = da.Patches(img, npatch, rel_overlap=overlap)
    out = da.PiecewisePerspectiveTransform().find_and_warp(patches, disp, reverse=rev)
    ref = W.find_and_warp(img, patches.rois, patches.local_corners_voxels, patches.global_corners_voxels, disp, rev)
    chk("warp", out.img, ref, (shape, np.dtype(dtype).name, npatch, overlap, ax, ay, rev))
    kw = dict(horizontal_bulge=float(rng.uniform(-3e-6, 3e-6)), vertical_bulge=float(rng.uniform(-3e-6, 3e-6)),
              horizontal_stretch=float(rng.uniform(-2e-6, 2e-6)), vertical_stretch=float(rng.uniform(-2e-6, 2e-6)),
              horizontal_center_offset=int(rng.integers(-5, 6)), vertical_center_offset=int(rng.integers(-5, 6)))
    cc = da.CurvatureCorrection(config={}, interpolation_order=int(rng.integers(0, 2)))
    chk("curvature", cc.simple_curvature_correction(img, **kw), CU.simple_curvature_correction(img, order=cc.interpolation_order, **kw),
        (shape, np.dtype(dtype).name, kw, cc.interpolation_order))
    if c == 0 and dtype != np.uint8:
        x0, rhs = rng.random(shape).astype(dtype), rng.random(shape).astype(dtype)
        mass = (0.5 + rng.random(shape)).astype(dtype) if rng.random() < 0.7 else 0.8
        diff = (0.2 + rng.random(shape)).astype(dtype) if (rng.random() < 0.7 or np.ndim(mass) == 0) else 0.4
        kwm = dict(depth=int(rng.integers(0, 2)), smoother_iterations=int(rng.integers(1, 4)), maxiter=int(rng.integers(1, 3)), dim=2)
        a = da.MG(mass_coeff=mass.copy() if np.ndim(mass) else mass, diffusion_coeff=diff.copy() if np.ndim(diff) else diff, **kwm)
        b = SO.MG(mass_coeff=mass.copy() if np.ndim(mass) else mass, diffusion_coeff=diff.copy() if np.ndim(diff) else diff, **kwm)
        o1, r1 = a(x0, rhs), b(x0, rhs)
        chk("mg het", o1, r1, (shape, np.dtype(dtype).name, kwm, np.ndim(mass), np.ndim(diff)))
        chk("mg het 2nd call", a(o1, rhs), b(r1, rhs), (shape, kwm))
        if dtype == np.float64:
            kc = dict(maxiter=400, tol=1e-10, dim=2, mass_coeff=0.9, diffusion_coeff=diff.copy() if np.ndim(diff) else diff)
            chk("cg", da.CG(**kc)(x0, rhs), SO.CG(**kc)(x0, rhs), (shape, np.ndim(diff)), tol=1e-8)
print(f"{n_cases} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
