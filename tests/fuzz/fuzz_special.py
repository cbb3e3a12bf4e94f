"""Developer fuzz: special values -- exact zeros, constants, tiny and huge magnitudes, negative values -- through the
K-iteration pipelines (reciprocal division with its IEEE fallback) against the oracle, bit for bit."""
import os, sys, warnings
os.environ.setdefault("DXH_PROMOTION", "native")   # these sweeps pin the native float32 kernels (oracle flow "strict")
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from oracle import tvd_numpy as O
warnings.simplefilter("ignore")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
n = 0
for dt, scales in ((np.float32, [1e-42, 1e-38, 1e-36, 1e-30, 1e-20, 1e-10, 1.0, 1e10, 1e20, 1e30, 1e36]),
                   (np.float64, [1e-320, 1e-305, 1e-300, 1e-200, 1e-100, 1.0, 1e100, 1e200, 1e300])):
    for sc in scales:
        for pattern in ("random", "signed", "zeros", "const", "sparse"):
            for shape in ((23, 300, 3), (40, 131)):
                base = rng.random(shape)
                if pattern == "signed": base = base - 0.5
                if pattern == "zeros": base[:] = 0.0
                if pattern == "const": base[:] = 0.37
                if pattern == "sparse": base[rng.random(shape) < 0.9] = 0.0
                with np.errstate(all="ignore"):
                    img = (base * sc).astype(dt)
                kws = [dict(mu=0.1, max_num_iter=8), dict(mu=0.1, omega=0.0, max_num_iter=5), dict(mu=1e-30, ell=1.0, max_num_iter=5)]
                # the stop test compares norms: NumPy squares in the array dtype (under/overflow), the device in double
                if (dt == np.float32 and 1e-15 < sc < 1e15) or (dt == np.float64 and 1e-120 < sc < 1e120):
                    kws.append(dict(mu=0.1, max_num_iter=9, eps=1e-3))
                for kw in kws:
                    with np.errstate(all="ignore"):
                        ref = O.split_bregman_tvd(img, promotion="strict", **kw)
                    out = da.split_bregman_tvd(img, **kw)
                    n += 1
                    same = np.array_equal(out, ref, equal_nan=True)
                    if not same:
                        bad += 1
                        fin = np.isfinite(ref) & np.isfinite(out)
                        d = np.abs(out[fin].astype(np.float64) - ref[fin].astype(np.float64)).max() if fin.any() else float("nan")
                        print("MISMATCH", np.dtype(dt).name, sc, pattern, shape, kw, "max abs(finite)", d,
                              "nan/inf ref", int((~np.isfinite(ref)).sum()), "out", int((~np.isfinite(out)).sum()), flush=True)
print(f"{n} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
