"""Developer fuzz for round 4's multi-sweep kernels against the NumPy oracle, bit for bit:
  (a) Jacobi(maxiter = 2 .. 7) in 2-D, one pass per iteration (k_sweeps_np2): float32 images in the default ("numpy2") flow with
      a scalar or float64-array mu, float64 images; with / without eps; random chunk counts;
  (b) volumes with Jacobi(maxiter = 2 .. 20) in plane tiles (k_jacobi_vol_multi): in-plane sizes from 16 x 32 up, float32 in
      both flows and float64, scalar or array-valued omega;
  (c) the Jacobi operator / H1_regularization on float64 operands (plain sweeps passes), 2-D and 3-D.
usage: python tests/fuzz/fuzz_r4.py [seed] [cases]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from oracle import tvd_numpy as O

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
t0 = time.time()


def check(tag, desc, out, ref):
    global bad
    if out.dtype != ref.dtype or not np.array_equal(out, ref, equal_nan=True):
        bad += 1
        d = np.abs(out.astype(np.float64) - ref.astype(np.float64))
        print("MISMATCH", tag, desc, out.dtype, ref.dtype, "max abs", float(np.nanmax(d)), "n", int(np.count_nonzero(d)), flush=True)


for case in range(n_cases):
    kind = case % 3
    os.environ.pop("DXH_NCHUNKS", None)
    if kind == 0:
        h, w = int(rng.integers(2, 420)), int(rng.integers(2, 700))
        c = int(rng.choice([0, 0, 1, 3, 3]))
        shape = (h, w) if c == 0 else (h, w, c)
        dt = np.float32 if rng.random() < 0.6 else np.float64
        img = (rng.random(shape) * float(rng.choice([1.0, 1.0, 255.0, 1e-3]))).astype(dt)
        S = int(rng.integers(2, 8))
        kw = dict(max_num_iter=int(rng.integers(1, 7)), omega=float(rng.choice([0.5, 1.0, 1.7])))
        sc = float(img.max()) or 1.0
        kw["mu"] = float(rng.choice([0.05, 0.2, 1.0])) * sc
        if rng.random() < 0.7:
            kw["ell"] = float(rng.choice([0.3, 1.0, 2.5])) * sc
        if dt == np.float32 and "ell" in kw and rng.random() < 0.4:
            kw["mu"] = kw["mu"] * (0.2 + 0.8 * rng.random(shape))          # float64 weight map: the threshold plane
            if rng.random() < 0.3:
                kw["mu"][rng.random(shape) < 0.03] *= -1.0
        if rng.random() < 0.3:
            kw["eps"] = float(rng.choice([3e-2, 5e-3]))
        if rng.random() < 0.5 and h >= 64:
            os.environ["DXH_NCHUNKS"] = str(int(rng.integers(1, 5)))
        promo = "numpy2" if dt == np.float32 else "strict"
        out = da.split_bregman_tvd(img, solver=da.Jacobi(maxiter=S), **kw)
        ref = O.split_bregman_tvd(img, jacobi_maxiter=S, promotion=promo, **kw)
        check("2-D sweeps", (shape, np.dtype(dt).name, S, {k: (v if np.ndim(v) == 0 else "array") for k, v in kw.items()}, os.environ.get("DXH_NCHUNKS")), out, ref)
    elif kind == 1:
        shape = (int(rng.integers(1, 40)), int(rng.integers(16, 70)), int(rng.integers(32, 140)))
        flow = rng.choice(["f32 numpy2", "f32 native", "f64"])
        dt = np.float64 if flow == "f64" else np.float32
        vol = rng.random(shape).astype(dt)
        S = int(rng.choice([2, 3, 4, 5, 6, 7, 9, 20]))
        kw = dict(mu=float(rng.choice([0.1, 0.3])), omega=float(rng.choice([0.6, 1.0])), ell=float(rng.choice([0.3, 0.8])), dim=3,
                  max_num_iter=int(rng.integers(1, 4)))
        if rng.random() < 0.35:
            kw["omega"] = (0.5 + rng.random(shape)).astype(np.float64 if flow == "f32 numpy2" else dt)
        if rng.random() < 0.3:
            kw["isotropic"] = flow != "f32 numpy2"
        if rng.random() < 0.25:
            kw["eps"] = 2e-2
        pk = {} if flow == "f64" else dict(promotion="numpy2" if flow == "f32 numpy2" else "native")
        out = da.split_bregman_tvd(vol, solver=da.Jacobi(maxiter=S), **pk, **kw)
        ref = O.split_bregman_tvd(vol, jacobi_maxiter=S, promotion="numpy2" if flow == "f32 numpy2" else "strict", **kw)
        check("volume sweeps", (shape, flow, S, {k: (v if np.ndim(v) == 0 else "array") for k, v in kw.items()}), out, ref)
    else:
        dim = int(rng.choice([2, 2, 3]))
        if dim == 2:
            h, w = int(rng.integers(2, 300)), int(rng.integers(2, 600))
            c = int(rng.choice([0, 1, 3]))
            shape = (h, w) if c == 0 else (h, w, c)
        else:
            shape = (int(rng.integers(1, 30)), int(rng.integers(16, 60)), int(rng.integers(32, 120)))
        x0, rhs = rng.random(shape), rng.random(shape)
        m = int(rng.integers(2, 14))
        mc, dc = float(rng.choice([0.3, 0.7, 1.5])), float(rng.choice([0.1, 0.3, 2.0]))
        out = da.Jacobi(maxiter=m, dim=dim, mass_coeff=mc, diffusion_coeff=dc)(x0, rhs)
        ref = O.jacobi(x0, rhs, mc, dc, dim=dim, maxiter=m, promotion="strict")
        check("Jacobi operator", (shape, dim, m, mc, dc), out, ref)
os.environ.pop("DXH_NCHUNKS", None)
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
