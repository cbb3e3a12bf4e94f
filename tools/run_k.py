"""Run the TV sweep with the current env knobs (for profiling a single configuration)."""
import os, sys
sys.path.insert(0, ".")
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 24
img = synthetic_image(4000, 8000, 3)
st = da.DeviceTVD(img.shape, img.dtype, dim=2)
st.set_coeffs(0.1, 1.0, None, img)
st.upload(img)
r = st.run(iters)
print("ms/iter", r.loop_ms / iters, "launches", r.launches)
