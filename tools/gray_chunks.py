"""Developer sweep: row-chunk count of the K = 4 gray float32 pipeline (4000 x 8000, 100 iterations), DXH_NCHUNKS per
process.  Usage: python tools/gray_chunks.py [nchunks [dtype]]   (0 = the cost model's choice; DXH_FUSED_TUNE=512 prints the geometry)"""
import os, sys
sys.path.insert(0, ".")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dtype = sys.argv[2] if len(sys.argv) > 2 else "float32"
if n:
    os.environ["DXH_NCHUNKS"] = str(n)
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
img = np.ascontiguousarray(synthetic_image(4000, 8000, 1).reshape(4000, 8000)).astype(dtype)
with da.DeviceTVD(img.shape, img.dtype) as st:
    st.set_coeffs(0.1, 1.0, None, img)
    best = 1e9
    for _ in range(5):
        st.upload(img)
        r = st.run(100)
        best = min(best, r.loop_ms)
    print(f"{dtype} nchunks={n or 'model'}: {best:.3f} ms per 100 iterations = {best / 25 * 1e3:.1f} us per launch", flush=True)
