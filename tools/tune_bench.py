"""Developer A/B of the fused sweep's tuning knobs inside ONE process (interleaved rounds).
usage: tune_bench.py "NAME:ENV=V,ENV=V" ...   (default set below)"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import darsia_amd as da
from darsia_amd.synthetic import synthetic_image

H, W, C = 4000, 8000, 3
iters = 48
img = synthetic_image(H, W, C)
default = ["k1w8:DXH_FUSE_K=1,DXH_WAVES_PER_CU=8", "k2w8:DXH_FUSE_K=2,DXH_WAVES_PER_CU=8",
           "k3w8:DXH_FUSE_K=3,DXH_WAVES_PER_CU=8", "k4w8:DXH_FUSE_K=4,DXH_WAVES_PER_CU=8",
           "k2w6:DXH_FUSE_K=2,DXH_WAVES_PER_CU=6", "k2w12:DXH_FUSE_K=2,DXH_WAVES_PER_CU=12",
           "k3w6:DXH_FUSE_K=3,DXH_WAVES_PER_CU=6", "k3w12:DXH_FUSE_K=3,DXH_WAVES_PER_CU=12",
           "k4w6:DXH_FUSE_K=4,DXH_WAVES_PER_CU=6", "k4w5:DXH_FUSE_K=4,DXH_WAVES_PER_CU=5"]
specs = sys.argv[1:] or default
ctxs = []
for spec in specs:
    name, envs = spec.split(":")
    for kv in envs.split(","):
        k, v = kv.split("=")
        os.environ[k] = v
    st = da.DeviceTVD(img.shape, img.dtype, dim=2)
    st.set_coeffs(0.1, 1.0, None, img)
    st.upload(img)
    st.run(12)
    ctxs.append((name, st, []))
for rnd in range(5):
    for name, st, ts in ctxs:
        r = st.run(iters)
        ts.append(r.loop_ms / iters)
npc = H * W * C
for name, st, ts in ctxs:
    med, mn = float(np.median(ts)), min(ts)
    print(f"{name:12s} median {med:.4f} ms/iter  min {mn:.4f}   {H*W/med/1e3:.0f} MPix/s", flush=True)
