"""Developer timing of the TV sweep variants on one GPU (not the judged bench)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import darsia_amd as da
from darsia_amd import _lib
from darsia_amd.synthetic import synthetic_image

H, W, C = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (4000, 8000, 3)))
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 100
t0 = time.time(); img = synthetic_image(H, W, C); print(f"gen {time.time()-t0:.1f}s", img.shape, img.dtype, flush=True)
shape = img.shape
with da.DeviceTVD(shape, img.dtype, dim=2) as st:
    st.set_coeffs(0.1, 1.0, None, img)
    t0 = time.time(); st.upload(img); print(f"upload {time.time()-t0:.2f}s", flush=True)
    for name, fl in (("fused_dpp", 0), ("fused_bpermute", _lib.FLAG_NO_DPP), ("three_kernel", _lib.FLAG_NO_FUSED)):
        st.run(3, flags=fl)
        best = None
        for rep in range(3):
            r = st.run(iters, flags=fl)
            ms = r.loop_ms / iters
            best = ms if best is None else min(best, ms)
        npc = H * W * C
        print(f"{name:16s} {best:.4f} ms/iter  {H*W/best/1e3:.0f} MPix/s  moved(7 vals) {7*4*npc/best/1e6:.0f} GB/s  "
              f"ref-44B-equiv {44*npc/best/1e6:.0f} GB/s", flush=True)
