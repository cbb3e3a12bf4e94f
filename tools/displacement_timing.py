"""Developer timing: DiffeomorphicImageRegistration.displacement() on a 4000 x 8000 frame, N_patches = [10, 20]."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.imageregistration import DiffeomorphicImageRegistration

class Table:
    def __init__(self): self.k = 0
    def find_effective_translation(self, *a, **k):
        self.k += 1
        return np.array([[1.0, 0.0, 3.0 * np.sin(self.k)], [0.0, 1.0, 2.0 * np.cos(self.k)]]), True

H, W = 4000, 8000
img = da.Image(np.random.default_rng(0).random((H, W)).astype(np.float32), dimensions=[1.0, 2.0])
reg = DiffeomorphicImageRegistration(img, N_patches=[10, 20], rel_overlap=0.1, translation_estimator=Table())
reg.translation_analysis.load_image(img)
t0 = time.perf_counter(); reg.translation_analysis.find_translation(); print("find_translation %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for _ in range(3):
    t0 = time.perf_counter(); d = reg.displacement(); dt = (time.perf_counter() - t0) * 1e3
    print("displacement() %.1f ms  shape %s  checksum %.6e" % (dt, d.shape, float(d[::100003].sum())), flush=True)
    del d
for _ in range(3):
    t0 = time.perf_counter(); w = reg.apply(img); dt = (time.perf_counter() - t0) * 1e3
    print("apply() (translate_image: find_and_warp with the spline pair as displacement) %.1f ms" % dt, flush=True)
    del w
