import sys, time, threading
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
img = synthetic_image(4000, 8000, 3)
N = img.size
for rep in range(3):
    t0 = time.perf_counter(); out = np.empty(N, np.float32); t1 = time.perf_counter()
    out[::1024] = 0; t2 = time.perf_counter()
    print(f"empty {1e3*(t1-t0):.2f} ms  touch {1e3*(t2-t1):.2f} ms", flush=True)
    del out
def prefault(a):
    a.reshape(-1)[::1024] = 0
for rep in range(3):
    st = da.DeviceTVD(img.shape, img.dtype, dim=2); st.set_coeffs(0.1, 1.0, None, img); st.upload(img)
    t0 = time.perf_counter()
    out = np.empty(img.shape, img.dtype)
    th = threading.Thread(target=prefault, args=(out,)); th.start()
    r = st.run(100)
    t1 = time.perf_counter()
    th.join(); t2 = time.perf_counter()
    from darsia_amd import _lib
    _lib.check(st.lib.dxh_tvd_download(st._h, _lib.ptr(out), None, None)); t3 = time.perf_counter()
    print(f"run {1e3*(t1-t0):.2f} join {1e3*(t2-t1):.2f} download {1e3*(t3-t2):.2f} total {1e3*(t3-t0):.2f}", flush=True)
    st.close()
