"""A few launches of one pipeline at the BASELINE size, for profiling (tools/sq_any.sh).
usage: run_variant.py aniso|iso|het|chambolle|warp|vol3d|gray|f64|np2|np2arr|jacobi3d"""
import sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
v = sys.argv[1] if len(sys.argv) > 1 else "aniso"
H, W = 4000, 8000
if v in ("aniso", "iso", "het"):
    img = synthetic_image(H, W, 3)
    mu = 0.1
    if v == "het":
        mu = np.full(img.shape, 0.1, np.float32); mu[:, W // 2:] = 0.3
    st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(mu, 1.0, None, img); st.upload(img)
    r = st.run(24, isotropic=(v == "iso")); print(v, "ms/iter", r.loop_ms / 24, "launches", r.launches); st.close()
elif v in ("gray", "f64"):
    img = synthetic_image(H, W, 1).astype(np.float64 if v == "f64" else np.float32)
    st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img)
    r = st.run(24); print(v, "ms/iter", r.loop_ms / 24); st.close()
elif v == "chambolle":
    img = synthetic_image(H, W, 3)
    out, it = da.denoise_tv_chambolle(img, weight=0.1, max_num_iter=12, eps=0.0, return_iters=True); print("chambolle iterations", it)
elif v == "chambolle60":
    img = synthetic_image(H, W, 3)
    da.denoise_tv_chambolle(img, weight=0.1, max_num_iter=60, eps=0.0)
elif v == "warp":
    img = da.Image(synthetic_image(H, W, 1)); patches = da.Patches(img, [10, 20])
    disp = lambda p: np.array([6 * np.sin(np.pi * p[:, 0] / W), 4 * np.sin(np.pi * p[:, 1] / H)])
    da.PiecewisePerspectiveTransform().find_and_warp(patches, disp)
elif v == "vol3d":
    vol = np.random.default_rng(0).random((256, 256, 256)).astype(np.float32)
    st = da.DeviceTVD(vol.shape, vol.dtype, dim=3); st.set_coeffs(0.1, 1.0, None, vol); st.upload(vol); st.run(8); st.close()
elif v in ("np2", "np2arr"):
    img = synthetic_image(H, W, 3)
    mu = 0.1 if v == "np2" else 0.1 * (0.3 + 0.7 * np.random.default_rng(1).random(img.shape))
    st = da.DeviceTVD(img.shape, img.dtype, promotion="numpy2"); st.set_coeffs(mu, 1.0, 0.2, img); st.upload(img)
    r = st.run(12); print(v, "ms/iter", r.loop_ms / 12, "launches", r.launches); st.close()
elif v == "np2gray":
    img = synthetic_image(H, W, 1)
    st = da.DeviceTVD(img.shape, img.dtype, promotion="numpy2"); st.set_coeffs(0.1, 1.0, 0.2, img); st.upload(img)
    r = st.run(12); print(v, "ms/iter", r.loop_ms / 12, "launches", r.launches); st.close()
elif v == "banded":
    img = synthetic_image(H, W, 3)
    out = da.split_bregman_tvd(img, mu=0.1, max_num_iter=40)     # the default call on a host image: row bands, numpy2 flow
    print("banded default call done", out.dtype, float(out[0, 0, 0]))
elif v == "jacobi3d":
    vol = np.random.default_rng(0).random((256, 256, 256)).astype(np.float32)
    st = da.DeviceTVD(vol.shape, vol.dtype, dim=3); st.set_coeffs(0.1, 1.0, None, vol); st.upload(vol)
    r = st.run(3, jacobi_maxiter=20); print(v, "ms/outer iteration", r.loop_ms / 3, "launches", r.launches); st.close()
