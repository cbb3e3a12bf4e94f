"""Developer check + timing of the S-sweeps-per-iteration numpy2 pipeline (dxh_np2_sweeps.h) against the three-launch flow
(DXH_FUSED_TUNE bit 33554432) and, on small images, the oracle."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image


def run(img, S, n, tune, chunks=None):
    os.environ["DXH_FUSED_TUNE"] = str(tune)
    if chunks: os.environ["DXH_NCHUNKS"] = str(chunks)
    else: os.environ.pop("DXH_NCHUNKS", None)
    with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
        st.set_coeffs(0.1, 1.0, 0.2, img); st.upload(img)
        r = st.run(n, jacobi_maxiter=S)
        r2 = st.run(2, jacobi_maxiter=S)   # a second call on the float64 iterate
        return st.download(), r.launches


bad = 0
rng = np.random.default_rng(0)
for shape in [(197, 263), (197, 263, 3), (64, 129, 3), (1000, 1531, 3), (1203, 2048)]:
    img = rng.random(shape).astype(np.float32)
    for S in (2, 3, 4, 5, 6):
        for chunks in (None, 3):
            if chunks and shape[0] < 900: continue
            a, la = run(img, S, 5, 0, chunks)
            b, lb = run(img, S, 5, 33554432)
            ok = np.array_equal(a, b)
            bad += not ok
            print(shape, "S", S, "chunks", chunks, "launches", la, lb, "equal" if ok else f"MISMATCH max {np.abs(a.astype(np.float64) - b).max():.3g} at {np.argwhere(a != b)[:3].tolist()}", flush=True)
if "--oracle" in sys.argv:
    from oracle import split_bregman_numpy as O
    print("oracle check in tests instead")
if "--time" in sys.argv:
    img = synthetic_image(4000, 8000, 3)
    for S in (2, 3, 5, 6):
        for tune in (0, 33554432):
            os.environ["DXH_FUSED_TUNE"] = str(tune)
            with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
                st.set_coeffs(0.1, 1.0, 0.2, img); st.upload(img); st.run(3, jacobi_maxiter=S)
                r = min((st.run(10, jacobi_maxiter=S) for _ in range(3)), key=lambda r: r.loop_ms)
                print(f"RGB Jacobi({S}) tune {tune}: {r.loop_ms / 10:.3f} ms/it ({r.launches} launches)", flush=True)
    g = np.ascontiguousarray(synthetic_image(4000, 8000, 3)[..., 0])
    for S in (5,):
        for tune in (0, 33554432):
            os.environ["DXH_FUSED_TUNE"] = str(tune)
            with da.DeviceTVD(g.shape, g.dtype, promotion="numpy2") as st:
                st.set_coeffs(0.1, 1.0, 0.2, g); st.upload(g); st.run(3, jacobi_maxiter=S)
                r = min((st.run(10, jacobi_maxiter=S) for _ in range(3)), key=lambda r: r.loop_ms)
                print(f"gray Jacobi({S}) tune {tune}: {r.loop_ms / 10:.3f} ms/it ({r.launches} launches)", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
