import sys, os
sys.path.insert(0, ".")
import numpy as np
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
for shape in ((4000, 8000, 3), (4000, 8000)):
    img = synthetic_image(shape[0], shape[1], 3 if len(shape) == 3 else 1)
    for iso in (True, False):
        for K in ("4", "3", "2"):
            os.environ["DXH_FUSE_K"] = K
            with da.DeviceTVD(img.shape, img.dtype) as st:
                st.set_coeffs(0.1, 1.0, None, img)
                st.upload(img)
                st.run(12, isotropic=iso)
                r = min((st.run(120, isotropic=iso) for _ in range(3)), key=lambda r: r.loop_ms)
                print(f"{shape} iso={iso} K={K}: {r.loop_ms / 120:.4f} ms/iteration ({r.launches} launches)", flush=True)
