"""Developer fuzz for round 3's kernels against the NumPy oracle, bit for bit:
  (a) the numpy2 promotion flow (k_fused_np2): random 2-D / colour float32 images, scalar or array mu (float32 / float64),
      scalar omega / ell, eps, special magnitudes -- against the oracle's reference ("numpy2") flow;
  (b) dim = 3 volumes (k_march3d_v, k_march3d, k_fused3d, k_jacobi_vol_x, k_rhs_vol, k_shrink_vol): random shapes whose rows
      are / are not 16-byte multiples, float32 / float64, anisotropic / isotropic / array mu, one or several Jacobi sweeps.
usage: python tests/fuzz/fuzz_r3.py [seed] [cases]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from oracle import tvd_numpy as O

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
t0 = time.time()


def report(tag, shape, dt, kw, out, ref):
    global bad
    bad += 1
    o64, r64 = out.astype(np.float64), ref.astype(np.float64)
    both = np.isfinite(o64) & np.isfinite(r64)
    print("MISMATCH", tag, shape, np.dtype(dt).name, {k: (v if np.ndim(v) == 0 else f"array({v.dtype})") for k, v in kw.items()},
          "max abs", float(np.abs(o64 - r64).max()), "| non-finite: device", int((~np.isfinite(o64)).sum()), "oracle", int((~np.isfinite(r64)).sum()),
          "| max abs where both finite", float(np.abs(o64 - r64)[both].max()) if both.any() else 0.0, flush=True)


for case in range(n_cases):
    if case % 2 == 0:
        # ---- (a) numpy2 flow
        h, w = int(rng.integers(1, 120)), int(rng.integers(1, 900))
        c = int(rng.choice([0, 0, 1, 2, 3, 4]))
        shape = (h, w) if c == 0 else (h, w, c)
        scale = float(rng.choice([1.0, 1.0, 1.0, 1e-6, 1e4, 255.0]))
        img = (rng.random(shape) * scale).astype(np.float32)
        if rng.random() < 0.2:
            img[rng.random(shape) < 0.3] = 0.0
        kw = dict(max_num_iter=int(rng.integers(1, 16)), ell=float(rng.choice([0.2, 0.5, 1.0, 3.0])), omega=float(rng.choice([0.5, 1.0, 2.0])))
        kind = rng.choice(["scalar", "f64", "f32"])
        if kind == "scalar":
            kw["mu"] = float(rng.choice([0.0, 0.05, 0.1, 0.3]))
            if rng.random() < 0.3:
                del kw["ell"]            # ell = 2 mu as a Python float
                if kw["mu"] == 0.0:
                    kw["mu"] = 0.1
        else:
            mu = 0.02 + 0.2 * rng.random(shape)
            if rng.random() < 0.5:
                mu[rng.random(shape) < 0.2] = 0.0
            kw["mu"] = mu if kind == "f64" else mu.astype(np.float32)
        if rng.random() < 0.3:
            kw["eps"] = float(rng.choice([1e-2, 3e-3]))
        if rng.random() < 0.25:
            # array-valued omega and / or ell (or ell = 2 mu of an array mu): float64 arrays throughout (k_rhs_np2_het)
            if np.ndim(kw["mu"]) != 0:
                kw["mu"] = np.maximum(kw["mu"].astype(np.float64), 0.01)
            which = rng.choice(["omega", "ell", "both", "ellnone"])
            if which in ("omega", "both"):
                kw["omega"] = 0.5 + rng.random(shape)
            if which in ("ell", "both"):
                kw["ell"] = 0.2 + 0.8 * rng.random(shape)
            if which == "ellnone":
                kw.pop("ell", None)
                if np.ndim(kw["mu"]) == 0:
                    kw["mu"] = 0.02 + 0.2 * rng.random(shape)
        # one sweep (the fused pipelines), several, or sweeps with a tolerance (three kinds of launches); now and then six channels
        jm = int(rng.choice([1, 1, 1, 2, 3, 5, 8]))
        tol = float(rng.choice([1e-2, 1e-3])) if (jm > 1 and rng.random() < 0.3) else None
        if rng.random() < 0.1 and c == 0:
            img = np.repeat(img[..., None], 6, axis=2) * rng.random((h, w, 6)).astype(np.float32)
            shape = img.shape
            for key in ("mu", "omega", "ell"):
                if key in kw and np.ndim(kw[key]) != 0:
                    kw[key] = np.repeat(kw[key][..., None], 6, axis=2)
        # now and then the adaptive schedule (scalar omega / ell, mu a scalar or a float64 array) or a warm start
        extra = {}
        scalars = np.ndim(kw["omega"]) == 0 and "ell" in kw and np.ndim(kw["ell"]) == 0
        if scalars and (np.ndim(kw["mu"]) == 0 or kw["mu"].dtype == np.float64) and rng.random() < 0.15:
            when = set(int(v) for v in rng.integers(0, kw["max_num_iter"], size=2))
            extra["adaptive"] = lambda it, when=when: it in when
        ref = O.split_bregman_tvd(img, jacobi_maxiter=jm, jacobi_tol=tol, **kw, **extra)   # the reference's NumPy-2 dtype flow
        out = da.split_bregman_tvd(img, promotion="numpy2", solver=da.Jacobi(maxiter=jm, tol=tol), **kw, **extra)
        if out.dtype != ref.dtype or not np.array_equal(out, ref, equal_nan=True):
            report(f"numpy2 jm={jm} tol={tol} adaptive={'adaptive' in extra}", shape, np.float32, kw, out, ref)
        if scalars and not extra and "eps" not in kw and rng.random() < 0.15:
            # warm start: continue from the state the oracle holds after the run above
            dim = 2
            _, dl, bl, _ = O.split_bregman_tvd(img, jacobi_maxiter=jm, jacobi_tol=tol, return_state=True, **kw)
            kw2 = dict(kw, max_num_iter=int(rng.integers(0, 4)))
            ref2, dl2, bl2, _ = O.split_bregman_tvd(img, jacobi_maxiter=jm, jacobi_tol=tol, return_state=True,
                                                    x0=(ref, [a.copy() for a in dl], [a.copy() for a in bl]), **kw2)
            d0, b0 = np.stack(dl, axis=-1), np.stack(bl, axis=-1)
            out2 = da.split_bregman_tvd(img, promotion="numpy2", solver=da.Jacobi(maxiter=jm, tol=tol), x0=(ref, d0, b0), **kw2)
            if not (np.array_equal(out2, ref2) and np.array_equal(d0, np.stack(dl2, axis=-1)) and np.array_equal(b0, np.stack(bl2, axis=-1))):
                dd = float(np.abs(d0.astype(np.float64) - np.stack(dl2, axis=-1)).max())
                bb = float(np.abs(b0.astype(np.float64) - np.stack(bl2, axis=-1)).max())
                report(f"numpy2 warm start jm={jm} tol={tol} scale={scale} |dd|={dd:.3e} |db|={bb:.3e} dtypes {d0.dtype} {dl2[0].dtype}",
                       shape, np.float32, kw2, out2, ref2)
        if case % 10 == 0:
            # Jacobi.__call__ (float64 out) and H1_regularization (float32 out) on the same image
            sol = da.Jacobi(maxiter=jm, tol=tol)
            sol.update_params(mass_coeff=kw["omega"] if np.ndim(kw["omega"]) == 0 else 0.8,
                              diffusion_coeff=kw["ell"] if ("ell" in kw and np.ndim(kw["ell"]) == 0) else 0.3, dim=2)
            rhs = (rng.random(img.shape) * scale).astype(np.float32)
            ref = O.jacobi(img, rhs, sol.mass_coeff, sol.diffusion_coeff, dim=2, maxiter=jm, tol=tol)
            out = sol(img, rhs)
            if out.dtype != ref.dtype or not np.array_equal(out, ref):
                report(f"jacobi numpy2 jm={jm} tol={tol}", shape, np.float32, {}, out, ref)
            ref = O.h1_regularization(img, 0.4, 1.3, dim=2, jacobi_maxiter=jm, jacobi_tol=tol)
            out = da.H1_regularization(img, 0.4, 1.3, dim=2, solver=da.Jacobi(maxiter=jm, tol=tol))
            if out.dtype != ref.dtype or not np.array_equal(out, ref):
                report(f"h1 numpy2 jm={jm} tol={tol}", shape, np.float32, {}, out, ref)
    else:
        # ---- (b) volumes
        n0, n1 = int(rng.integers(1, 40)), int(rng.integers(1, 60))
        n2 = int(rng.choice([rng.integers(1, 300), 4 * rng.integers(1, 80), 64, 68, 128]))
        shape = (n0, n1, n2)
        dt = rng.choice([np.float32, np.float64])
        vol = rng.random(shape).astype(dt)
        kw = dict(max_num_iter=int(rng.integers(1, 7)), dim=3)
        het = rng.random() < 0.3
        kw["mu"] = (0.02 + 0.2 * rng.random(shape)).astype(dt) if het else float(rng.choice([0.05, 0.1, 0.3]))
        if rng.random() < 0.3:
            kw["omega"] = float(rng.choice([0.5, 2.0]))
        if rng.random() < 0.3:
            kw["ell"] = float(rng.choice([0.3, 1.0]))
        iso = rng.random() < 0.3
        jm = int(rng.choice([1, 1, 2, 3, 20]))
        # the native kernels (vector march, sweeps) against the oracle's same-dtype flow ...
        ref = O.split_bregman_tvd(vol, isotropic=iso, jacobi_maxiter=jm, promotion="strict", **kw)
        out = da.split_bregman_tvd(vol, isotropic=iso, solver=da.Jacobi(maxiter=jm), promotion="native", **kw)
        if not np.array_equal(out, ref):
            report(f"dim3 iso={iso} jm={jm}", shape, dt, kw, out, ref)
        # ... and, where the reference's own float32 flow exists (anisotropic, scalar ell), the DEFAULT call against it
        if dt == np.float32 and not iso and (not het or "ell" in kw):
            kw2 = dict(kw)
            if het:
                kw2["mu"] = kw["mu"].astype(np.float64) if rng.random() < 0.5 else kw["mu"]
            ref = O.split_bregman_tvd(vol, jacobi_maxiter=jm, promotion="numpy2", **kw2)
            out = da.split_bregman_tvd(vol, solver=da.Jacobi(maxiter=jm), **kw2)
            if not np.array_equal(out, ref):
                report(f"dim3 numpy2 jm={jm}", shape, dt, kw2, out, ref)
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
