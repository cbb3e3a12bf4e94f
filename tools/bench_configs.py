"""Timings of the BASELINE.json configs other than the headline one (wall time of whole calls; run under
rocprofv3 --kernel-trace --stats for kernel-only durations)."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
from scipy.interpolate import RBFInterpolator

def t(label, f, n=1):
    f(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    dt = (time.perf_counter() - t0) / n
    print(f"{label:70s} {dt*1e3:10.2f} ms", flush=True)
    return r

def loop_ms(st, n, **kw):
    """ms per iteration of the device loop: one warm-up run of the same length (clock ramp), best of three."""
    st.run(n, **kw)
    return min(st.run(n, **kw).loop_ms for _ in range(3)) / n

H, W = 4000, 8000
rgb = synthetic_image(H, W, 3)
gray = synthetic_image(H, W, 1)
print("config 2 whole call (upload + 100 it + download), f32 RGB")
t("split_bregman_tvd(rgb, mu=.1, 100 it)  [PCIe-inclusive]", lambda: da.split_bregman_tvd(rgb, mu=0.1, max_num_iter=100))
st = da.DeviceTVD(rgb.shape, rgb.dtype); st.set_coeffs(0.1, 1.0, None, rgb); st.upload(rgb)
for iso in (False, True):
    ms = loop_ms(st, 100, isotropic=iso)
    print(f"  device loop {'isotropic' if iso else 'anisotropic'}: {ms:.4f} ms/iter  {H*W/ms/1e3:.0f} MPix/s")
st.close()
g64 = gray.astype(np.float64)
st = da.DeviceTVD(g64.shape, g64.dtype); st.set_coeffs(0.1, 1.0, None, g64); st.upload(g64); ms = loop_ms(st, 100)
print(f"  device loop float64 gray: {ms:.4f} ms/iter  {H*W/ms/1e3:.0f} MPix/s"); st.close()
st = da.DeviceTVD(gray.shape, gray.dtype); st.set_coeffs(0.1, 1.0, None, gray); st.upload(gray); ms = loop_ms(st, 100)
print(f"  device loop float32 gray: {ms:.4f} ms/iter  {H*W/ms/1e3:.0f} MPix/s"); st.close()
print("config 3: colour apply + isotropic TVD")
A = np.eye(3) + 0.05 * np.random.default_rng(7).normal(size=(3, 3)); b = 0.01 * np.random.default_rng(7).normal(size=3)
t("apply_affine_balance(rgb f32)  [PCIe-inclusive]", lambda: da.apply_affine_balance(rgb, A, b, clip=True))
t("split_bregman_tvd(rgb, isotropic, 100 it) [PCIe-inclusive]", lambda: da.split_bregman_tvd(rgb, mu=0.1, max_num_iter=100, isotropic=True))
u8 = (rgb * 255.0 + 0.5).astype(np.uint8)
def chain_device():
    d = da.DeviceArray.from_host(u8)
    out = da.split_bregman_tvd(da.apply_affine_balance(d, A, b), mu=0.1, max_num_iter=100, isotropic=True)
    return out.get()
def chain_host():
    return da.split_bregman_tvd(da.apply_affine_balance(u8, A, b), mu=0.1, max_num_iter=100, isotropic=True)
t("config 3 chain, device-resident (uint8 up, colour, 100 it isotropic, float32 down)", chain_device)
t("config 3 chain through host arrays between the operators", chain_host)
print("default TVD(): chambolle on RGB-as-volume, eps 2e-4")
out, iters = da.denoise_tv_chambolle(rgb, weight=0.1, return_iters=True)
del out   # dropping a 384 MB result costs the host ~20 ms (munmap): keep it out of the timed call
t0 = time.perf_counter(); out, iters = da.denoise_tv_chambolle(rgb, weight=0.1, return_iters=True); dt = time.perf_counter() - t0
print(f"  chambolle: {iters} iterations, {dt*1e3:.1f} ms total, {dt/max(iters,1)*1e3:.2f} ms/iteration [PCIe-inclusive]")
print("config 4: registration dense parts on two 4000 x 8000 gray frames, N_patches = [10, 20]")
img = da.Image(gray)
patches = da.Patches(img, [10, 20])
disp = lambda p: np.array([6 * np.sin(np.pi * p[:, 0] / W), 4 * np.sin(np.pi * p[:, 1] / H)])
t("PiecewisePerspectiveTransform.find_and_warp  [PCIe-inclusive]", lambda: da.PiecewisePerspectiveTransform().find_and_warp(patches, disp))
rng = np.random.default_rng(3)
c = rng.random((262, 2)) * np.array([W, H]); rbf = RBFInterpolator(c, rng.normal(size=262))
t("evaluate_rbf_on_grid (262 centres, 32 M pixels)  [incl. 256 MB download]", lambda: da.evaluate_rbf_on_grid(rbf, (H, W)))
print("ConcentrationAnalysis (RGB frame -> concentration), fused on the device vs NumPy stage chain on the host")
base_img = da.OpticalImage(rgb)
probe_img = da.OpticalImage(np.clip(rgb + 0.1 * np.random.default_rng(1).random(rgb.shape).astype(np.float32), 0, 1))
tvd_op = da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=100, eps=None)
model = da.CombinedModel([da.LinearModel(scaling=2.0, offset=0.0), da.ClipModel(min_value=0.0, max_value=1.0)])
ca = da.ConcentrationAnalysis(base_img, da.MonochromaticReduction(color="red"), None, tvd_op, model, **{"diff option": "positive"})
t("ConcentrationAnalysis.__call__ fused (upload RGB, 100 TVD it, download scalar)", lambda: ca(probe_img))
def series():
    return list(ca.map([probe_img] * 8))   # results kept (dropping a 128 MB array costs the host ~7 ms)
import time as _t
_r = series(); del _r; _t0 = _t.perf_counter(); _r = series(); _dt = (_t.perf_counter() - _t0) * 1e3 / 8
print(f"  ConcentrationAnalysis.map over 8 frames (upload of frame i+1 overlaps frame i): {_dt:.2f} ms/frame")
del _r
ca_ch = da.ConcentrationAnalysis(base_img, da.MonochromaticReduction(color="red"), None, da.TVD(), model, **{"diff option": "positive"})
t("ConcentrationAnalysis.__call__ with the default TVD() (chambolle) inside the fused pipeline", lambda: ca_ch(probe_img))
porosity = 0.2 + 0.6 * np.random.default_rng(5).random((H, W))          # float64 map, as the workflow's image_porosity
ca_por = da.ConcentrationAnalysis(base_img, da.MonochromaticReduction(color="red"), None,
                                  da.TVD(method="heterogeneous bregman", weight=0.1 * porosity, regularization=0.2, max_num_iter=100, eps=None),
                                  model, **{"diff option": "positive"})
t("ConcentrationAnalysis.__call__ with a porosity-weighted TVD (float64 weight map, 100 it)", lambda: ca_por(probe_img))
def host_chain():
    d = np.clip(probe_img.img - base_img.img, 0, None); s = d[:, :, 0]; m = np.clip(2.0 * s + 0.0, 0.0, 1.0)
    return m
t("  the same elementwise stages as NumPy passes on the host (no TVD)", host_chain)
print("heterogeneous coefficients: array-valued mu (device loop)")
mu = np.full(rgb.shape, 0.1, np.float32); mu[:, W // 2:] = 0.3
st = da.DeviceTVD(rgb.shape, rgb.dtype); st.set_coeffs(mu, 1.0, None, rgb); st.upload(rgb); ms = loop_ms(st, 99)
print(f"  device loop mu array (ell = 2 mu, array as well): {ms:.4f} ms/iter  {H*W/ms/1e3:.0f} MPix/s"); st.close()
st = da.DeviceTVD(rgb.shape, rgb.dtype); st.set_coeffs(mu, 1.0, 0.2, rgb); st.upload(rgb); ms = loop_ms(st, 99)
print(f"  device loop mu array, scalar omega and ell (the TVD operator / porosity weights): {ms:.4f} ms/iter  {H*W/ms/1e3:.0f} MPix/s"); st.close()
os.environ["DXH_FUSED_TUNE"] = "65536"
st = da.DeviceTVD(rgb.shape, rgb.dtype); st.set_coeffs(mu, 1.0, 0.2, rgb); st.upload(rgb); ms = loop_ms(st, 99)
print(f"    the same through the general array pipeline (DXH_FUSED_TUNE=65536): {ms:.4f} ms/iter"); st.close()
del os.environ["DXH_FUSED_TUNE"]
print("Gauss-Seidel split Bregman (TVD '* bregman' methods), 64 sweeps, eps off")
t("denoise_tv_bregman(rgb, 64 sweeps)  [PCIe-inclusive]", lambda: da.denoise_tv_bregman(rgb, weight=1.0, max_num_iter=64, eps=0.0, isotropic=False))
print("MG inside the TV loop (gray f32, depth 3, 2 smoother sweeps, 1 V-cycle per outer iteration, 20 outer iterations)")
g2 = np.ascontiguousarray(gray.reshape(H, W))
t("split_bregman_tvd(gray, solver=MG)  [PCIe-inclusive]", lambda: da.split_bregman_tvd(g2, mu=0.1, max_num_iter=20, solver=da.MG(depth=3, smoother_iterations=2, maxiter=1)))
print("dim = 3 volume 256^3 float32 (device loop)")
vol = np.random.default_rng(0).random((256, 256, 256)).astype(np.float32)
st = da.DeviceTVD(vol.shape, vol.dtype, dim=3); st.set_coeffs(0.1, 1.0, None, vol); st.upload(vol); ms = loop_ms(st, 40)
print(f"  fused 3-D sweep: {ms:.4f} ms/iter  {vol.size/ms/1e6:.1f} GVox/s"); st.close()
t("H1_regularization(rgb, mu=0.1, Jacobi(maxiter=20))  [PCIe-inclusive]", lambda: da.H1_regularization(rgb, mu=0.1, omega=1.0, solver=da.Jacobi(maxiter=20)))
print("Resize / CurvatureCorrection")
t("Resize(fx=fy=0.5, inter_area)(rgb)  [PCIe-inclusive]", lambda: da.Resize(fx=0.5, fy=0.5, interpolation="inter_area")(rgb))
t("Resize(shape=(3000, 6000), inter_linear)(rgb)  [PCIe-inclusive]", lambda: da.Resize(shape=(3000, 6000), interpolation="inter_linear")(rgb))
cc = da.CurvatureCorrection(config={"bulge": dict(horizontal_bulge=2e-9, vertical_bulge=-1e-9), "stretch": dict(horizontal_stretch=1e-9)})
cc(rgb)
t("CurvatureCorrection(rgb) with the cached grid  [PCIe-inclusive]", lambda: cc(rgb))
