"""Developer timing: pieces of split_bregman_tvd on a DeviceArray (isotropic, 100 iterations, 4000 x 8000 x 3)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
rgb = np.random.default_rng(0).random((4000, 8000, 3)).astype(np.float32)
d = da.DeviceArray.from_host(rgb)
def T(): return time.perf_counter()
for rep in range(3):
    t0 = T(); st = da.DeviceTVD(rgb.shape, rgb.dtype, dim=2); t1 = T()
    st.set_coeffs(0.1, 1.0, None, rgb); t2 = T()
    st.upload(d); t3 = T()
    r = st.run(100, isotropic=True); t4 = T()
    out = st.download(out=da.DeviceArray(rgb.shape, rgb.dtype)); t5 = T()
    st.close(); t6 = T()
    ms = [(y - x) * 1e3 for x, y in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6))]
    print("create %.2f  set_coeffs %.2f  upload %.2f  run %.2f (loop %.2f)  download %.2f  close %.2f ms" %
          (ms[0], ms[1], ms[2], ms[3], r.loop_ms, ms[4], ms[5]))
    out.close()
