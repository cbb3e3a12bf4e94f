"""Developer A/B of the fused sweep's tuning knobs inside ONE process (interleaved rounds)."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image

H, W, C = 4000, 8000, 3
iters = 50
img = synthetic_image(H, W, C)
configs = [("base tight/plain/noremap w16", dict(DXH_FUSED_TUNE="7", DXH_WAVES_PER_CU="16")),
           ("align+nts+remap w8", dict(DXH_FUSED_TUNE="0", DXH_WAVES_PER_CU="8")),
           ("align+nts+remap w4", dict(DXH_FUSED_TUNE="0", DXH_WAVES_PER_CU="4")),
           ("align+nts+remap w12", dict(DXH_FUSED_TUNE="0", DXH_WAVES_PER_CU="12")),
           ("align+nts+remap w16", dict(DXH_FUSED_TUNE="0", DXH_WAVES_PER_CU="16")),
           ("align+plain+remap w8", dict(DXH_FUSED_TUNE="2", DXH_WAVES_PER_CU="8")),
           ("align+nts+noremap w8", dict(DXH_FUSED_TUNE="4", DXH_WAVES_PER_CU="8")),
           ("tight+nts+remap w8", dict(DXH_FUSED_TUNE="1", DXH_WAVES_PER_CU="8")),
           ]
ctxs = []
for name, env in configs:
    os.environ.update(env)
    st = da.DeviceTVD(img.shape, img.dtype, dim=2)
    st.set_coeffs(0.1, 1.0, None, img)
    st.upload(img)
    st.run(3)
    ctxs.append((name, st, []))
for rnd in range(5):
    for name, st, ts in ctxs:
        r = st.run(iters)
        ts.append(r.loop_ms / iters)
npc = H * W * C
for name, st, ts in ctxs:
    med, mn = float(np.median(ts)), min(ts)
    print(f"{name:32s} median {med:.4f} ms  min {mn:.4f} ms   {H*W/med/1e3:.0f} MPix/s   {28*npc/med/1e6:.0f} GB/s (28B)", flush=True)
