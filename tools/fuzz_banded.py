"""Developer fuzz: dxh_tvd_solve_banded against the one-piece device run: random shapes, band counts, iteration counts,
dtypes, Jacobi sweeps, isotropic, scalar / array coefficients (incl. arrays of the other float type)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.restoration.split_bregman_tvd import solve_banded

def one_piece(img, mu, omega, ell, n, m, iso):
    with da.DeviceTVD(img.shape, img.dtype, dim=2) as st:
        st.set_coeffs(mu, omega, ell, img); st.upload(img); st.run(n, m, None, eps=None, isotropic=iso); return st.download()

def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    bad = 0
    for i in range(n_cases):
        h = int(rng.integers(9, 700)); w = int(rng.integers(8, 500)); c = int(rng.choice([0, 1, 2, 3, 4]))
        shape = (h, w) if c == 0 else (h, w, c)
        dt = np.float32 if rng.random() < 0.6 else np.float64
        img = rng.random(shape).astype(dt)
        n = int(rng.integers(1, 40)); m = int(rng.choice([1, 1, 1, 2, 3])); iso = bool(rng.random() < 0.3)
        nb = int(rng.integers(0, 9))
        def coef(lo, hi):
            r = rng.random()
            if r < 0.6: return float(lo + (hi - lo) * rng.random())
            a = lo + (hi - lo) * rng.random(shape)
            return a.astype(dt if rng.random() < 0.7 else (np.float64 if dt == np.float32 else np.float32))
        mu, omega = coef(0.02, 0.3), coef(0.5, 2.0)
        ell = None if rng.random() < 0.5 else coef(0.1, 0.6)
        cast = lambda v: v.astype(dt) if isinstance(v, np.ndarray) else v
        want = one_piece(img, cast(mu), cast(omega), None if ell is None else cast(ell), n, m, iso)
        got, used = solve_banded(img, mu, omega, ell, n, m, iso, nbands=nb, return_bands=True)
        ok = np.array_equal(got, want)
        kind = lambda v: "arr:" + v.dtype.name if isinstance(v, np.ndarray) else "scalar"
        print(("ok      " if ok else "MISMATCH"), shape, dt.__name__, f"n={n} m={m} iso={iso} nbands={nb}->{used} mu={kind(mu)} omega={kind(omega)} ell={'none' if ell is None else kind(ell)}", flush=True)
        bad += int(not ok)
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
