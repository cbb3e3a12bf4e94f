"""Developer fuzz: Chambolle, Gauss-Seidel Bregman, Jacobi, MG, map_coordinates, resize against their oracles."""
import os, sys, time
os.environ.setdefault("DXH_PROMOTION", "native")   # these sweeps pin the native float32 kernels (oracle flow "strict")
sys.path.insert(0, ".")
import numpy as np
from scipy.ndimage import map_coordinates as sp_map
import darsia_amd as da
from darsia_amd import resizing as rz
from oracle import bregman_gs_numpy as BG, chambolle_numpy as CH, resize_numpy as R, solvers_numpy as SO, tvd_numpy as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
def chk(name, a, b, info):
    global bad
    if a.shape != b.shape or not np.array_equal(a, b):
        bad += 1
        print("MISMATCH", name, info, flush=True)
t0 = time.time()
for case in range(n_cases):
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 400))
    c = int(rng.choice([0, 1, 3]))
    shape = (h, w) if c == 0 else (h, w, c)
    dt = rng.choice([np.float32, np.float64])
    img = rng.random(shape).astype(dt)
    it = int(rng.integers(1, 12)); eps = float(rng.choice([0.0, 1e-3, 1e-2])); wgt = float(rng.choice([0.05, 0.2, 1.0]))
    o, n = da.denoise_tv_chambolle(img, weight=wgt, eps=eps, max_num_iter=it, return_iters=True)
    r, nr = CH.denoise_tv_chambolle(img, weight=wgt, eps=eps, max_num_iter=it, return_iters=True)
    chk("chambolle", o, r, (shape, dt.__name__, wgt, eps, it, n, nr)); bad += int(n != nr)
    iso = bool(rng.random() < 0.5)
    o, n = da.denoise_tv_bregman(img, weight=wgt * 5, max_num_iter=it, eps=eps, isotropic=iso, return_iters=True)
    r, nr = BG.denoise_tv_bregman(img, weight=wgt * 5, max_num_iter=it, eps=eps, isotropic=iso, return_iterations=True)
    chk("bregman_gs", o, r, (shape, dt.__name__, wgt, eps, it, iso, n, nr)); bad += int(n != nr)
    if c == 0 and h >= 8 and w >= 8:
        x0, rhs = rng.random(shape).astype(dt), rng.random(shape).astype(dt)
        kw = dict(depth=int(rng.integers(0, 2)), smoother_iterations=int(rng.integers(1, 6)), maxiter=int(rng.integers(1, 3)), mass_coeff=1.0, diffusion_coeff=0.4, dim=2)
        chk("mg", da.MG(**kw)(x0, rhs), SO.MG(**kw)(x0, rhs), (shape, dt.__name__, kw))
    for src in (img, (img * 255).astype(np.uint8), (img * 65535).astype(np.uint16)):
        order = int(rng.integers(0, 2))
        cdt = rng.choice([np.float32, np.float64])
        coords = np.stack([rng.uniform(-1, h, 500), rng.uniform(-1, w, 500)]).astype(cdt)
        coords[:, :4] = np.array([[0, h - 1, 0, h - 1], [0, w - 1, w - 1, 0]], dtype=cdt)
        got = da.map_coordinates(src, coords, order=order)
        ref = np.stack([sp_map(src[..., k], coords, order=order) for k in range(src.shape[2])], axis=-1) if src.ndim == 3 else sp_map(src, coords, order=order)
        chk("map_coordinates", got, ref, (shape, src.dtype.name, order, np.dtype(cdt).name))
    if h >= 2 and w >= 2:
        interp = int(rng.choice([R.INTER_NEAREST, R.INTER_LINEAR, R.INTER_AREA]))
        fx, fy = float(rng.uniform(0.3, 1.7)), float(rng.uniform(0.3, 1.7))
        if round(w * fx) >= 1 and round(h * fy) >= 1:
            o = rz.cv_resize(img, fx=fx, fy=fy, interpolation=interp)
            chans = [np.ascontiguousarray(img[..., k]) for k in range(img.shape[2])] if img.ndim == 3 else [img]
            r = [R.resize(ch, fx=fx, fy=fy, interpolation=interp) for ch in chans]
            r = np.stack(r, axis=-1) if img.ndim == 3 else r[0]
            chk("resize", o, r, (shape, dt.__name__, interp, fx, fy))
            u8 = (img * 255).astype(np.uint8)
            o = rz.cv_resize(u8, fx=fx, fy=fy, interpolation=interp)
            chans = [np.ascontiguousarray(u8[..., k]) for k in range(u8.shape[2])] if u8.ndim == 3 else [u8]
            r = [R.resize(ch, fx=fx, fy=fy, interpolation=interp) for ch in chans]
            r = np.stack(r, axis=-1) if u8.ndim == 3 else r[0]
            chk("resize u8", o, r, (shape, interp, fx, fy))
            dw, dh = int(rng.integers(1, 2 * w + 2)), int(rng.integers(1, 2 * h + 2))
            o = rz.cv_resize(img, dsize=(dw, dh), interpolation=interp)
            chans = [np.ascontiguousarray(img[..., k]) for k in range(img.shape[2])] if img.ndim == 3 else [img]
            r = [R.resize(ch, dsize=(dw, dh), interpolation=interp) for ch in chans]
            r = np.stack(r, axis=-1) if img.ndim == 3 else r[0]
            chk("resize dsize", o, r, (shape, dt.__name__, interp, dw, dh))
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
