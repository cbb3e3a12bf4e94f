"""Developer timing: 3-D volumes (dim = 3) through the three-kernel path."""
import sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
rng = np.random.default_rng(0)
for shape, dt in (((256, 256, 256), np.float32), ((256, 256, 256), np.float64), ((200, 300, 180), np.float32)):
    vol = rng.random(shape).astype(dt)
    for jm in (1, 20):
        st = da.DeviceTVD(vol.shape, vol.dtype, dim=3); st.set_coeffs(0.1, 1.0, None, vol); st.upload(vol)
        st.run(3, jacobi_maxiter=jm)
        ts = [st.run(20, jacobi_maxiter=jm) for _ in range(2)]
        ms = min(t.loop_ms for t in ts) / 20
        n = vol.size
        print(f"{shape} {np.dtype(dt).name} Jacobi(maxiter={jm}): {ms:.3f} ms/iter  {n/ms/1e6:.1f} GVox/s  launches/iter {ts[0].launches/20:.0f}", flush=True)
        st.close()
    if dt == np.float32:
        # the reference example's coefficients (examples/regularization.py:14-29): mu an array, omega = 0.1, ell = 1.0
        mu = np.ones(shape, dt); mu[:, : shape[1] // 3, :] = 0.5
        st = da.DeviceTVD(vol.shape, vol.dtype, dim=3); st.set_coeffs(mu, 0.1, 1.0, vol); st.upload(vol)
        st.run(3, jacobi_maxiter=20)
        ms = min(st.run(10, jacobi_maxiter=20).loop_ms for _ in range(2)) / 10
        print(f"{shape} {np.dtype(dt).name} array-valued mu, Jacobi(maxiter=20): {ms:.3f} ms/iter", flush=True)
        st.run(3)
        ms = min(st.run(20).loop_ms for _ in range(2)) / 20
        print(f"{shape} {np.dtype(dt).name} array-valued mu, Jacobi(maxiter=1): {ms:.3f} ms/iter", flush=True)
        st.close()
        # the reference's own float32 dtype flow (promotion "numpy2": float64 context, three kinds of launches)
        for jm in (1, 20):
            st = da.DeviceTVD(vol.shape, vol.dtype, dim=3, promotion="numpy2"); st.set_coeffs(0.1, 1.0, None, vol); st.upload(vol)
            st.run(3, jacobi_maxiter=jm)
            ms = min(st.run(10, jacobi_maxiter=jm).loop_ms for _ in range(2)) / 10
            print(f"{shape} float32 numpy2 flow, Jacobi(maxiter={jm}): {ms:.3f} ms/iter", flush=True)
            st.close()
