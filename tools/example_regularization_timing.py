"""Developer timing: the calls of the reference's examples/regularization.py (a volume with a heterogeneous weight, anisotropic
TVD with Jacobi(maxiter=20) and eps; H1 regularisation with Jacobi(maxiter=30, tol=1e-6)) on a synthetic 192^3 volume,
and the TVD call of examples/paper/pet_simulations_comparison_block_b.py (isotropic, array omega, 100 iterations, eps)."""
import io, sys, time, contextlib
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
rng = np.random.default_rng(0)
n = 192
vol = rng.random((n, n, n))
mu = np.ones_like(vol); mu[:, 0:100, :] = 0.5
img = da.ScalarImage(vol, dimensions=[0.1, 0.1, 0.1], space_dim=3) if hasattr(da, "ScalarImage") else vol
def timed(label, f):
    for rep in range(2):
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            out = f()
        dt = (time.perf_counter() - t0) * 1e3
    print(f"{label}: {dt:.1f} ms", flush=True)
    return out
for dt in (np.float64, np.float32):
    v = vol.astype(dt)
    timed(f"{np.dtype(dt).name} tvd heterogeneous bregman, weight map, Jacobi(20), eps=1e-6, 30 iterations", lambda: da.split_bregman_tvd(v, mu=mu, omega=0.1, max_num_iter=30, eps=1e-6, dim=3, solver=da.Jacobi(maxiter=20)))
    timed(f"{np.dtype(dt).name} ... verbose=True", lambda: da.split_bregman_tvd(v, mu=mu, omega=0.1, max_num_iter=30, eps=1e-6, dim=3, verbose=True, solver=da.Jacobi(maxiter=20)))
    timed(f"{np.dtype(dt).name} H1_regularization Jacobi(30, tol=1e-6)", lambda: da.H1_regularization(v, mu=1, omega=1, dim=3, solver=da.Jacobi(maxiter=30, tol=1e-6)))
    om = 0.01 + 0.02 * rng.random(v.shape).astype(dt)
    if dt == np.float64:
        timed(f"{np.dtype(dt).name} tvd isotropic, array omega, Jacobi(20), eps=1e-5, 100 iterations", lambda: da.split_bregman_tvd(v, mu=0.02, omega=om, isotropic=True, max_num_iter=100, eps=1e-5, dim=3, solver=da.Jacobi(maxiter=20)))
