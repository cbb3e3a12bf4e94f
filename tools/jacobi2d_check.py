"""Developer check + timing: multi-sweep Jacobi passes on float64 contexts (2-D) through the LDS-row sweeps pipeline in its PLAIN
form (dxh_np2_sweeps.h) against k_jacobi_multi<double> (DXH_FUSED_TUNE bit 134217728)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image

def solve(x0, rhs, m, tune):
    os.environ["DXH_FUSED_TUNE"] = str(tune)
    sol = da.Jacobi(maxiter=m, mass_coeff=0.7, diffusion_coeff=0.3)
    return sol(x0, rhs)

bad = 0
rng = np.random.default_rng(0)
for shape in [(197, 263), (197, 263, 3), (64, 129, 3), (1000, 1531, 3), (1203, 2048), (30, 40, 1)]:
    x0, rhs = rng.random(shape), rng.random(shape)
    for m in (2, 3, 4, 5, 6, 7, 9, 11, 20):
        a = solve(x0, rhs, m, 0)
        b = solve(x0, rhs, m, 134217728)
        ok = np.array_equal(a, b)
        bad += not ok
        print(shape, "m", m, "equal" if ok else f"MISMATCH max {np.abs(a - b).max():.3g} n={np.count_nonzero(a != b)}", flush=True)
    img = rng.random(shape)
    for m in (8, 12):   # the TV loop with more than six sweeps, float64 image and float32 image (numpy2 flow)
        for dt in (np.float64, np.float32):
            outs = []
            for tune in (0, 134217728):
                os.environ["DXH_FUSED_TUNE"] = str(tune)
                outs.append(da.split_bregman_tvd(img.astype(dt), mu=0.1, ell=0.3, max_num_iter=3, solver=da.Jacobi(maxiter=m)))
            ok = np.array_equal(*outs)
            bad += not ok
            print(shape, np.dtype(dt).name, "TV loop m", m, "equal" if ok else "MISMATCH", flush=True)
if "--time" in sys.argv:
    import time
    for C in (3, 1):
        img = synthetic_image(4000, 8000, 3).astype(np.float64)
        if C == 1: img = np.ascontiguousarray(img[..., 0])
        for tune in (0, 134217728):
            os.environ["DXH_FUSED_TUNE"] = str(tune)
            with da.DeviceTVD(img.shape, img.dtype) as st:
                st.set_coeffs(0.1, 1.0, 0.2, img); st.upload(img); st.run(2, jacobi_maxiter=20)
                r = min((st.run(5, jacobi_maxiter=20) for _ in range(3)), key=lambda r: r.loop_ms)
                print(f"float64 C={C} TV loop Jacobi(20) tune {tune}: {r.loop_ms / 5:.3f} ms/it ({r.launches} launches)", flush=True)
            d = da.DeviceArray.from_host(img)
            for _ in range(2):
                t0 = time.perf_counter()
                out = da.H1_regularization(d, mu=0.4, omega=1.3, solver=da.Jacobi(maxiter=30))
                out.get() if False else None
                dt_ = (time.perf_counter() - t0) * 1e3
            print(f"float64 C={C} H1_regularization(device array, Jacobi(30)) tune {tune}: {dt_:.2f} ms", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
