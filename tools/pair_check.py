"""Developer check: wave-pair K=4 kernel (DXH_FUSED_TUNE=16) against the solo K=1 path, bit for bit."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image

ok = True
for shape in [(64, 96, 3), (37, 53, 3), (200, 1000, 3), (129, 257), (513, 700, 3), (1000, 2000, 3)]:
    img = synthetic_image(*shape) if len(shape) == 3 else np.ascontiguousarray(synthetic_image(shape[0], shape[1], 3)[..., 0])
    outs = {}
    for name, env in (("k1", {"DXH_FUSE_K": "1", "DXH_FUSED_TUNE": "0"}), ("pair", {"DXH_FUSE_K": "4", "DXH_FUSED_TUNE": "16"})):
        os.environ.update(env)
        for iso in (False, True):
            st = da.DeviceTVD(img.shape, img.dtype, dim=2)
            st.set_coeffs(0.1, 1.0, None, img)
            st.upload(img)
            st.run(12, isotropic=iso)
            outs[(name, iso)] = st.download()
    for iso in (False, True):
        same = np.array_equal(outs[("k1", iso)], outs[("pair", iso)])
        print(shape, "iso" if iso else "aniso", "bit-exact" if same else "MISMATCH %g" % np.abs(outs[("k1", iso)] - outs[("pair", iso)]).max(), flush=True)
        ok = ok and same
sys.exit(0 if ok else 1)
