"""Developer scan: how many float divisors does the exhaustive check reject (three-operation division not exact)?"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from darsia_amd import _lib
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(0)
ds = rng.uniform(1, 2, 2 * n).astype(np.float32)
bad = C.c_uint(0); rej = []
for a, b in zip(ds[::2], ds[1::2]):
    _lib.check(lib.dxh_div3_probe(0, float(a), 0.0, float(b), 0.0, C.byref(bad)))
    if bad.value & 1: rej.append(float(a))
    if bad.value & 2: rej.append(float(b))
print(f"{2*n} divisors, {len(rej)} rejected", [repr(x) for x in rej[:8]])
