import sys, time, ctypes as C
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd import _lib
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(4000, 8000, 3)
lib = _lib.load()
da.denoise_tv_chambolle(img[:64, :64], max_num_iter=2)
def call(n, out):
    done = C.c_int(0)
    t = time.perf_counter()
    _lib.check(lib.dxh_chambolle(0, 0, 3, 4000, 8000, 3, _lib.ptr(img), _lib.ptr(out), 0.1, 0.0, n, C.byref(done)))
    return (time.perf_counter() - t) * 1e3
out = np.empty_like(img)
for n in (0, 1, 2, 10, 44):
    print(n, "its: fresh out %.1f ms" % call(n, np.empty_like(img)), " reused out %.1f ms" % call(n, out), flush=True)
