import os, sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
H = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
img = synthetic_image(H, 8000, 3)
for nch in (0, 32, 64, 128, 256):
    if nch: os.environ["DXH_NCHUNKS"] = str(nch)
    st = da.DeviceTVD(img.shape, img.dtype, dim=2); st.set_coeffs(0.1, 1.0, None, img); st.upload(img)
    st.run(8); r = min(st.run(40).loop_ms for _ in range(2)) / 40
    print(f"H={H} nchunks={nch or 'default'}: {r:.4f} ms/iter  ({H*8000/r/1e3:.0f} MPix/s)", flush=True)
    st.close()
