"""Developer check + timing: K Jacobi sweeps of a volume per pass (dxh_vol_multi.h) against sweep by sweep
(DXH_FUSED_TUNE bit 67108864), numpy2 and native flows, float32 and float64 volumes."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da

def run(vol, m, n, tune, promotion, omega=1.0):
    os.environ["DXH_FUSED_TUNE"] = str(tune)
    with da.DeviceTVD(vol.shape, vol.dtype, dim=3, promotion=promotion) as st:
        st.set_coeffs(0.1, omega, 0.3, vol); st.upload(vol)
        r = st.run(n, jacobi_maxiter=m)
        return st.download(), r.launches

bad = 0
rng = np.random.default_rng(0)
for shape in [(20, 33, 70), (40, 64, 64), (17, 100, 131), (64, 16, 32)]:
    for dt, promo in ((np.float32, "numpy2"), (np.float32, "native"), (np.float64, "native")):
        vol = rng.random(shape).astype(dt)
        om_arr = (0.5 + rng.random(shape)).astype(np.float64 if promo == "numpy2" else dt)   # array-valued omega: a divisor plane
        for m, om in [(m, 1.0) for m in (2, 3, 4, 5, 7, 20)] + [(m, om_arr) for m in (2, 3, 6, 20)]:
            a, la = run(vol, m, 3, 0, promo, om)
            b, lb = run(vol, m, 3, 67108864, promo, om)
            ok = np.array_equal(a, b)
            bad += not ok
            print(shape, np.dtype(dt).name, promo, "m", m, "omega", "array" if om is om_arr else "scalar", "launches", la, lb, "equal" if ok else f"MISMATCH {np.abs(a.astype(np.float64)-b).max():.3g} n={np.count_nonzero(a != b)}", flush=True)
if "--time" in sys.argv:
    for n in (256, 384):
        for dt, promo in ((np.float32, "numpy2"), (np.float64, "native"), (np.float32, "native")):
            vol = rng.random((n, n, n)).astype(dt)
            om_arr = (0.5 + rng.random(vol.shape)).astype(np.float64 if promo == "numpy2" else dt)
            for om, name in ((1.0, "scalar omega"), (om_arr, "array omega")):
                if n > 256 and om is om_arr: continue
                for tune in (0, 67108864):
                    os.environ["DXH_FUSED_TUNE"] = str(tune)
                    with da.DeviceTVD(vol.shape, vol.dtype, dim=3, promotion=promo) as st:
                        st.set_coeffs(0.1, om, None, vol); st.upload(vol); st.run(2, jacobi_maxiter=20)
                        r = min((st.run(5, jacobi_maxiter=20) for _ in range(3)), key=lambda r: r.loop_ms)
                        print(f"{n}^3 {np.dtype(dt).name} {promo} {name} Jacobi(20) tune {tune}: {r.loop_ms / 5:.3f} ms/it ({r.launches} launches)", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
