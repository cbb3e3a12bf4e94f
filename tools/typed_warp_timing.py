"""Developer timing: find_and_warp on 4000 x 8000 frames in the type they are stored in (10 x 20 patches); "through float32 on
the host" is the path before: astype(float32), float32 over PCIe both ways, astype back."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.registration import warp_patches

H, W = 4000, 8000
rng = np.random.default_rng(0)


def disp(p):
    p = np.asarray(p, dtype=float)
    return np.array([6.0 * np.sin(np.pi * p[:, 0] / W), 4.0 * np.sin(np.pi * p[:, 1] / H)])


def best(f, n=4):
    t, keep = 1e9, []   # the results stay alive while timing: dropping a 128 - 768 MB array costs the host milliseconds of munmap
    for _ in range(n):
        t0 = time.perf_counter(); keep.append(f()); t = min(t, (time.perf_counter() - t0) * 1e3)
    return t


pw = da.PiecewisePerspectiveTransform()
for shape in ((H, W), (H, W, 3)):
    for dtype in (np.float32, np.uint8, np.uint16, np.float64):
        img = rng.random(shape, dtype=np.float32)
        img = (img * np.iinfo(dtype).max).astype(dtype) if np.dtype(dtype).kind == "u" else img.astype(dtype)
        patches = da.Patches(da.Image(img), [10, 20], rel_overlap=0.0)
        t_new = best(lambda: pw.find_and_warp(patches, disp))
        line = f"{str(shape):16s} {np.dtype(dtype).name:8s} find_and_warp {t_new:7.1f} ms"
        if dtype != np.float32:
            f32 = da.Patches(da.Image(img.astype(np.float32)), [10, 20], rel_overlap=0.0)
            t_old = best(lambda: (img.astype(np.float32), pw.find_and_warp(f32, disp).img.astype(dtype))[1])
            line += f"   through float32 on the host {t_old:7.1f} ms"
        print(line, flush=True)
