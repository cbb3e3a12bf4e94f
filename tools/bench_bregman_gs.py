"""Developer timing: Gauss-Seidel split Bregman (wavefront sweeps) at full size."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4000, 8000)
img = synthetic_image(H, W, 3)
da.denoise_tv_bregman(img[:64, :64], weight=1.0, max_num_iter=1, eps=0.0)
for n in (1, 16, 64):
    t = time.perf_counter()
    out, it = da.denoise_tv_bregman(img, weight=1.0, max_num_iter=n, eps=0.0, isotropic=False, return_iters=True)
    dt = time.perf_counter() - t
    print(f"{H}x{W}x3 sweeps={it} wall {dt:.3f} s", flush=True)
