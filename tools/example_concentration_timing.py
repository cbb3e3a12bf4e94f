"""Developer timing: ConcentrationAnalysis as the reference's examples configure it (examples/co2_analysis.py:20-32: TVD() --
Chambolle; examples/kernel_interpolation.py:33-40: TVD(weight=0.025, eps=1e-4, max_num_iter=100, method="isotropic Bregman")),
4000 x 8000 x 3 frames, per call and in a series."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
H, W = 4000, 8000
rng = np.random.default_rng(0)
for dt in (np.float32, np.float64):
    rgb = rng.random((H, W, 3))
    frames = (rgb * 255).astype(np.uint8) if dt == np.uint8 else rgb.astype(dt)
    probe = np.clip(rgb + 0.1 * rng.random(rgb.shape), 0, 1)
    probe = (probe * 255).astype(np.uint8) if dt == np.uint8 else probe.astype(dt)
    base_img, probe_img = da.OpticalImage(frames), da.OpticalImage(probe)
    for name, tvd in (("TVD() (Chambolle)", da.TVD()), ("isotropic Bregman", da.TVD(weight=0.025, eps=1e-4, max_num_iter=100, method="isotropic Bregman"))):
        model = da.CombinedModel([da.LinearModel(scaling=4.0), da.ClipModel(min_value=0.0, max_value=1.0)])
        ca = da.ConcentrationAnalysis(base_img, da.MonochromaticReduction(color="red"), None, tvd, model)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ca(probe_img)
            t0 = time.perf_counter(); out = [ca(probe_img) for _ in range(3)]; per_call = (time.perf_counter() - t0) * 1e3 / 3
            t0 = time.perf_counter(); n = sum(1 for _ in ca.map([probe_img] * 6)); series = (time.perf_counter() - t0) * 1e3 / 6
        print(f"{np.dtype(dt).name} frames, {name}: {per_call:.1f} ms per call, {series:.1f} ms per frame in a series (path: {getattr(ca, 'last_path', '?')})", flush=True)
        ca.close()
