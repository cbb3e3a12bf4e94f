"""Developer fuzz, part 2: 3-D volumes (dim = 3), warm starts, H1, MG / CG inside the TV loop, adaptive schedule."""
import os, sys, time
os.environ.setdefault("DXH_PROMOTION", "native")   # these sweeps pin the native float32 kernels (oracle flow "strict")
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from oracle import solvers_numpy as SO, tvd_numpy as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
def chk(name, a, b, info, exact=True):
    global bad
    ok = np.array_equal(a, b) if exact else np.allclose(a, b, rtol=1e-5, atol=1e-6)
    if not ok:
        bad += 1
        print("MISMATCH", name, info, "max abs", float(np.abs(a.astype(float) - b.astype(float)).max()), flush=True)
t0 = time.time()
for case in range(n_cases):
    dt = rng.choice([np.float32, np.float64])
    kind = rng.choice(["vol", "warm", "h1", "mg", "adaptive"])
    if kind == "vol":
        shape = tuple(int(v) for v in rng.integers(1, 24, 3))
        vol = rng.random(shape).astype(dt)
        het = rng.random() < 0.4
        mu = (0.02 + 0.2 * rng.random(shape)).astype(dt) if het else 0.1
        jm = int(rng.choice([1, 2, 4])); n = int(rng.integers(1, 7)); iso = bool(rng.random() < 0.3)
        out = da.split_bregman_tvd(vol, mu=mu, dim=3, max_num_iter=n, isotropic=iso, solver=da.Jacobi(maxiter=jm))
        ref = O.split_bregman_tvd(vol, mu=mu, dim=3, max_num_iter=n, isotropic=iso, jacobi_maxiter=jm, promotion="strict")
        chk("vol", out, ref, (shape, dt.__name__, het, jm, n, iso), exact=not iso)
    elif kind == "warm":
        h, w = int(rng.integers(2, 50)), int(rng.integers(2, 300)); c = int(rng.choice([0, 3]))
        shape = (h, w) if c == 0 else (h, w, c)
        img = rng.random(shape).astype(dt)
        n1, n2 = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        d = np.zeros(shape + (2,), dt); b = np.zeros(shape + (2,), dt)
        first = da.split_bregman_tvd(img, mu=0.1, max_num_iter=n1, x0=(img.copy(), d, b))
        second = da.split_bregman_tvd(img, mu=0.1, max_num_iter=n2, x0=(first, d, b))
        chk("warm", second, O.split_bregman_tvd(img, mu=0.1, max_num_iter=n1 + n2, promotion="strict"), (shape, dt.__name__, n1, n2))
    elif kind == "h1":
        h, w = int(rng.integers(1, 50)), int(rng.integers(1, 300)); c = int(rng.choice([0, 3]))
        shape = (h, w) if c == 0 else (h, w, c)
        img = rng.random(shape).astype(dt); jm = int(rng.integers(1, 9))
        chk("h1", da.H1_regularization(img, mu=0.3, omega=0.8, dim=2, solver=da.Jacobi(maxiter=jm)),
            O.h1_regularization(img, mu=0.3, omega=0.8, dim=2, jacobi_maxiter=jm, promotion="strict"), (shape, dt.__name__, jm))
    elif kind == "mg":
        h, w = int(rng.integers(8, 60)), int(rng.integers(8, 200))
        img = rng.random((h, w)).astype(dt)
        sv = dict(depth=int(rng.integers(0, 3)), smoother_iterations=int(rng.integers(1, 5)), maxiter=int(rng.integers(1, 3)))
        if min(h, w) >> (sv["depth"] + 1) < 1:
            continue
        n = int(rng.integers(1, 5))
        chk("tvd+mg", da.split_bregman_tvd(img, mu=0.1, max_num_iter=n, solver=da.MG(**sv)),
            O.split_bregman_tvd(img, mu=0.1, max_num_iter=n, promotion="strict", solver=SO.MG(**sv)), ((h, w), dt.__name__, sv, n))
    else:
        h, w = int(rng.integers(2, 40)), int(rng.integers(2, 200))
        img = rng.random((h, w)).astype(np.float64)
        sched = (lambda i: i % 2 == 1)
        n = int(rng.integers(2, 7))
        chk("adaptive", da.split_bregman_tvd(img, mu=0.1, max_num_iter=n, adaptive=sched),
            O.split_bregman_tvd(img, mu=0.1, max_num_iter=n, promotion="strict", adaptive=sched), ((h, w), n))
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
