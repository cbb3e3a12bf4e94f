"""Developer timing: DeviceArray.get() of 384 MB into a fresh array vs into a reused one."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
a = np.random.default_rng(0).random((4000, 8000, 3)).astype(np.float32)
d = da.DeviceArray.from_host(a)
for rep in range(3):
    t0 = time.perf_counter(); out = d.get(); t1 = time.perf_counter()
    out2 = d.get(out=out); t2 = time.perf_counter()
    print(f"fresh {1e3*(t1-t0):.1f} ms   reused {1e3*(t2-t1):.1f} ms   equal {np.array_equal(out, a)}")
    del out, out2
