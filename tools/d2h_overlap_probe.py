"""Developer probe: does a device-to-host copy overlap a concurrent host-to-device copy of pageable memory -- into a touched
pageable destination vs a page-locked one?"""
import sys, time, threading, ctypes as C
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd import _lib
lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
n_up, n_dn = 384 << 20, 128 << 20
src = np.random.default_rng(0).integers(0, 255, n_up, dtype=np.uint8)
d_up = da.DeviceArray((n_up,), np.uint8)
d_dn = da.DeviceArray((n_dn,), np.uint8)
pageable = np.zeros(n_dn, np.uint8)
p = C.c_void_p()
assert hip.hipHostMalloc(C.byref(p), C.c_size_t(n_dn), 0) == 0
pinned = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n_dn,))
pinned[:] = 0
dev = d_up.device
def up():
    _lib.check(lib.dxh_mem_copy(dev, C.c_void_p(d_up.ptr), _lib.ptr(src), n_up))
def down(dst):
    _lib.check(lib.dxh_mem_copy(dev, _lib.ptr(dst), C.c_void_p(d_dn.ptr), n_dn))
for _ in range(2): up(); down(pageable); down(pinned)
def t(f, *a):
    t0 = time.perf_counter(); f(*a); return (time.perf_counter() - t0) * 1e3
print("alone: up %.2f  down pageable %.2f  down pinned %.2f" % (t(up), t(down, pageable), t(down, pinned)))
for name, dst in (("pageable", pageable), ("pinned", pinned)) * 2:
    res = {}
    th = threading.Thread(target=lambda: res.setdefault("up", [t(up) for _ in range(4)]))
    t0 = time.perf_counter(); th.start()
    time.sleep(0.001)
    downs = [t(down, dst) for _ in range(4)]
    th.join()
    print(f"concurrent, destination {name}: uploads {['%.2f' % v for v in res['up']]}  downloads {['%.2f' % v for v in downs]}  all {(time.perf_counter()-t0)*1e3:.1f} ms")

# the same with TV iterations running on a third thread's stream
img = np.random.default_rng(1).random((4000, 8000)).astype(np.float32)
stop = False
def compute():
    with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
        st.set_coeffs(0.1, 1.0, None, img); st.upload(img)
        while not stop:
            st.run(100)
tc = threading.Thread(target=compute); tc.start(); time.sleep(1.0)
for name, dst in (("pageable", pageable), ("pinned", pinned)) * 2:
    res = {}
    th = threading.Thread(target=lambda: res.setdefault("up", [t(up) for _ in range(4)]))
    t0 = time.perf_counter(); th.start()
    time.sleep(0.001)
    downs = [t(down, dst) for _ in range(4)]
    th.join()
    print(f"beside TV iterations, destination {name}: uploads {['%.2f' % v for v in res['up']]}  downloads {['%.2f' % v for v in downs]}  all {(time.perf_counter()-t0)*1e3:.1f} ms")
stop = True; tc.join()

# hipHostRegister of an ordinary (touched) NumPy buffer: cost, and copies into it beside uploads
reg = np.zeros(n_dn, np.uint8)
t0 = time.perf_counter(); rc = hip.hipHostRegister(C.c_void_p(reg.ctypes.data), C.c_size_t(n_dn), 0); t_reg = (time.perf_counter() - t0) * 1e3
print("hipHostRegister(128 MB touched) rc %d: %.2f ms" % (rc, t_reg))
reg2 = np.empty(n_dn, np.uint8)
t0 = time.perf_counter(); rc = hip.hipHostRegister(C.c_void_p(reg2.ctypes.data), C.c_size_t(n_dn), 0); t_reg = (time.perf_counter() - t0) * 1e3
print("hipHostRegister(128 MB fresh) rc %d: %.2f ms" % (rc, t_reg))
down(reg); down(reg)
print("alone: down registered %.2f" % t(down, reg))
for name, dst in (("registered", reg), ("a DIFFERENT pageable buffer each time", None)) * 2:
    res = {}
    th = threading.Thread(target=lambda: res.setdefault("up", [t(up) for _ in range(4)]))
    others = [np.zeros(n_dn, np.uint8) for _ in range(4)]
    t0 = time.perf_counter(); th.start()
    time.sleep(0.001)
    downs = [t(down, dst if dst is not None else others[i]) for i in range(4)]
    th.join()
    print(f"concurrent, destination {name}: uploads {['%.2f' % v for v in res['up']]}  downloads {['%.2f' % v for v in downs]}  all {(time.perf_counter()-t0)*1e3:.1f} ms")
t0 = time.perf_counter(); rc = hip.hipHostUnregister(C.c_void_p(reg.ctypes.data)); print("hipHostUnregister rc %d: %.2f ms" % (rc, (time.perf_counter() - t0) * 1e3))
