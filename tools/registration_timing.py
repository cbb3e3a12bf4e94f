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


class HostEstimator:
    """The estimator as it ran before (per patch, NumPy FFTs on the host), for the comparison only."""

    def find_effective_translation(self, img_src, img_dst, roi_src=None, roi_dst=None, mask_src=None, mask_dst=None, plot_matches=False):
        a, b = np.asarray(img_src, dtype=np.float64), np.asarray(img_dst, dtype=np.float64)
        wy, wx = np.hanning(a.shape[0])[:, None], np.hanning(a.shape[1])[None, :]
        fa = np.fft.rfft2((a - a.mean()) * wy * wx)
        fb = np.fft.rfft2((b - b.mean()) * wy * wx)
        cross = fb * np.conj(fa)
        cross /= np.maximum(np.abs(cross), 1e-12)
        corr = np.fft.irfft2(cross, s=a.shape)
        py, px = np.unravel_index(np.argmax(corr), corr.shape)
        h, w = corr.shape
        ref = lambda m, c, p: 0.0 if (m - 2 * c + p) == 0 else 0.5 * (m - p) / (m - 2 * c + p)
        dy = py + ref(corr[(py - 1) % h, px], corr[py, px], corr[(py + 1) % h, px])
        dx = px + ref(corr[py, (px - 1) % w], corr[py, px], corr[py, (px + 1) % w])
        dy, dx = (dy - h if dy > h / 2 else dy), (dx - w if dx > w / 2 else dx)
        return np.array([[1.0, 0.0, dx], [0.0, 1.0, dy]]), True


host = da.ImageRegistration(da.Image(dst), N_patches=[10, 20], rel_overlap=0.1, translation_estimator=HostEstimator())
t0 = time.perf_counter(); res = host(da.Image(src)); t1 = time.perf_counter()
print(f"with the per-patch NumPy estimator on the host (the path before): whole ImageRegistration.__call__ {1e3 * (t1 - t0):7.1f} ms", flush=True)
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
    reg_d = da.ImageRegistration(da.Image(da.DeviceArray.from_host(b.img)), N_patches=[10, 20], rel_overlap=0.1)
    a_d = da.Image(da.DeviceArray.from_host(a.img))
    for rep in range(3):
        t0 = time.perf_counter(); keep.append(reg_d(a_d)); t1 = time.perf_counter()
        print(f"{np.dtype(dtype).name}: both frames already in HBM (DeviceArray pixels): whole ImageRegistration.__call__ {1e3 * (t1 - t0):7.1f} ms", flush=True)
    assert np.array_equal(keep[-1].img.get(), keep[-4].img)
    multi = da.ImageRegistration(b, N_patches=[[5, 10], [10, 20]], rel_overlap=0.1)
    for rep in range(3):
        t0 = time.perf_counter(); keep.append(multi(a)); t1 = time.perf_counter()
        print(f"{np.dtype(dtype).name}: multilevel [[5, 10], [10, 20]]: whole ImageRegistration.__call__ {1e3 * (t1 - t0):7.1f} ms", flush=True)
    out = keep[-7].img.astype(np.float64) / (255 if dtype == np.uint8 else 1)
    inner = (slice(200, H - 200), slice(200, W - 200))
    print(f"   mean |src - dst| {np.abs(src - dst)[inner].mean():.5f} -> after registration {np.abs(out - dst)[inner].mean():.5f}")
