"""Developer check + timing: float64 images with Jacobi(maxiter = 2 .. 6) through the sweeps-per-iteration pipeline
(dxh_np2_sweeps.h, F = double) against the three launches per iteration (DXH_FUSED_TUNE bit 33554432)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image

def run(img, S, n, tune, chunks=None):
    os.environ["DXH_FUSED_TUNE"] = str(tune)
    if chunks: os.environ["DXH_NCHUNKS"] = str(chunks)
    else: os.environ.pop("DXH_NCHUNKS", None)
    with da.DeviceTVD(img.shape, img.dtype) as st:
        st.set_coeffs(0.1, 0.9, 0.3, img); st.upload(img)
        r = st.run(n, jacobi_maxiter=S)
        r2 = st.run(2, jacobi_maxiter=S)
        return st.download(), r.launches

bad = 0
rng = np.random.default_rng(0)
for shape in [(197, 263), (197, 263, 3), (64, 129, 3), (1000, 1531, 3), (1203, 2048)]:
    img = rng.random(shape)
    for S in (2, 3, 4, 5, 6):
        for chunks in (None, 3):
            if chunks and shape[0] < 900: continue
            a, la = run(img, S, 5, 0, chunks)
            b, lb = run(img, S, 5, 33554432)
            ok = np.array_equal(a, b)
            bad += not ok
            print(shape, "S", S, "chunks", chunks, "launches", la, lb, "equal" if ok else f"MISMATCH max {np.abs(a - b).max():.3g} n={np.count_nonzero(a != b)}", flush=True)
if "--time" in sys.argv:
    for C in (3, 1):
        img = synthetic_image(4000, 8000, 3).astype(np.float64)
        if C == 1: img = np.ascontiguousarray(img[..., 0])
        for S in (2, 3, 5):
            for tune in (0, 33554432):
                os.environ["DXH_FUSED_TUNE"] = str(tune)
                with da.DeviceTVD(img.shape, img.dtype) as st:
                    st.set_coeffs(0.1, 1.0, 0.2, img); st.upload(img); st.run(3, jacobi_maxiter=S)
                    r = min((st.run(10, jacobi_maxiter=S) for _ in range(3)), key=lambda r: r.loop_ms)
                    print(f"float64 C={C} Jacobi({S}) tune {tune}: {r.loop_ms / 10:.3f} ms/it ({r.launches} launches)", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
