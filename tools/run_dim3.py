"""A few dim = 3 iterations for profiling (tools/sq_any.sh)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
vol = np.random.default_rng(0).random((n, n, n)).astype(np.float32)
st = da.DeviceTVD(vol.shape, vol.dtype, dim=3); st.set_coeffs(0.1, 1.0, None, vol); st.upload(vol)
st.run(6)
st.close()
