"""Developer timing: the whole registration of configs[3] (two 4000 x 8000 gray frames, N_patches = [10, 20]) by stage."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from scipy.ndimage import gaussian_filter, map_coordinates

H, W = 4000, 8000
rng = np.random.default_rng(1234)
dst = gaussian_filter(rng.random((H, W)).astype(np.float32), 3.0)
dst = ((dst - dst.min()) / (dst.max() - dst.min())).astype(np.float32)
yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
src = map_coordinates(dst, [yy + 4.0 * np.sin(np.pi * yy / H), xx + 6.0 * np.sin(np.pi * xx / W)], order=1, mode="nearest").astype(np.float32)
for dtype in (np.float32, np.uint8):
    a = da.Image(src if dtype == np.float32 else (src * 255).astype(np.uint8))
    b = da.Image(dst if dtype == np.float32 else (dst * 255).astype(np.uint8))
    reg = da.ImageRegistration(b, N_patches=[10, 20], rel_overlap=0.1)
    ta = reg.image_registration.translation_analysis
    keep = []
    for rep in range(3):
        t0 = time.perf_counter(); ta.load_image(a); t1 = time.perf_counter()
        ta.find_translation(); t2 = time.perf_counter()
        keep.append(ta.translate_image(reverse=False)); t3 = time.perf_counter()
        keep.append(reg(a)); t4 = time.perf_counter()
        print(f"{np.dtype(dtype).name}: find_translation {1e3 * (t2 - t1):7.1f} ms   translate_image {1e3 * (t3 - t2):6.1f} ms   "
              f"whole ImageRegistration.__call__ {1e3 * (t4 - t3):7.1f} ms", flush=True)
    out = keep[-1].img.astype(np.float64) / (255 if dtype == np.uint8 else 1)
    inner = (slice(200, H - 200), slice(200, W - 200))
    print(f"   mean |src - dst| {np.abs(src - dst)[inner].mean():.5f} -> after registration {np.abs(out - dst)[inner].mean():.5f}")
