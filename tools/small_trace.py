import sys
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
img = np.random.default_rng(0).random((512, 512)).astype(np.float32)
st = da.DeviceTVD(img.shape, img.dtype); st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(20)
r = st.run(40); st.close()
print(r.loop_ms, r.launches)
