#!/usr/bin/env python3
"""Single-GPU timing of the multi-GPU schedule: a periodic slab (the rank is its own RCCL neighbour) of the size one
rank owns in configs[4] at N = 2 / 4 / 8, with the overlapped interface-first order, with serialised exchanges, and a
plain context of the same rows without any exchange.  The copy goes through RCCL's local path instead of xGMI, so
this shows launch-order / stream overheads, not link time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import darsia_amd as da
from darsia_amd import _lib
from darsia_amd.parallel import Comm, SlabTVD, make_plan, periodic_slab
from darsia_amd.synthetic import synthetic_image

W, C, K_HALO, ITERS, STEPS = 16000, 3, int(os.environ.get("HALO", "20")), 100, 3
os.environ.setdefault("MASTER_PORT", "29611")
comm = Comm.rccl(0, 1, device=0)
# warm the clocks: the first timed configuration otherwise reads a few per cent slow
_img = synthetic_image(2000, W, C)
with da.DeviceTVD(_img.shape, _img.dtype, dim=2) as _st:
    _st.set_coeffs(0.1, 1.0, None, _img); _st.upload(_img); _st.run(400)
for n in (8, 4, 2):
    rows = 8000 // n
    img = synthetic_image(rows, W, C)
    line = {"ranks": n, "rows": rows}
    for mode in ("overlap", "serial"):
        plan = make_plan(rows, 1, 0, K_HALO, periodic=True)
        s = SlabTVD(plan, periodic_slab(img, plan), mu=0.1, halo_iters=K_HALO, comm=comm, promotion=os.environ.get('DXH_BENCH_FLOW', 'native'),
                    flags=_lib.FLAG_NO_OVERLAP if mode == "serial" else 0)
        s.run(ITERS, more_to_come=True, asynchronous=True)
        s.synchronize()
        best = 1e9
        for _rep in range(3):
            t0 = time.perf_counter()
            for _ in range(STEPS):
                s.run(ITERS, more_to_come=True, asynchronous=True)
            s.synchronize()
            best = min(best, (time.perf_counter() - t0) / STEPS * 1e3)
        line[mode + "_ms_per_step"] = round(best, 3)
        s.close()
    with da.DeviceTVD(img.shape, img.dtype, dim=2) as st:
        st.set_coeffs(0.1, 1.0, None, img)
        st.upload(img)
        st.run(ITERS)
        best = 1e9
        for _rep in range(3):
            t0 = time.perf_counter()
            for _ in range(STEPS):
                st.run(ITERS)
            best = min(best, (time.perf_counter() - t0) / STEPS * 1e3)
        line["no_exchange_ms_per_step"] = round(best, 3)
    print(line, flush=True)
# the whole image on one GPU, for the ratio
img = synthetic_image(8000, W, C)
with da.DeviceTVD(img.shape, img.dtype, dim=2) as st:
    st.set_coeffs(0.1, 1.0, None, img); st.upload(img); st.run(ITERS)
    best = 1e9
    for _rep in range(3):
        t0 = time.perf_counter(); st.run(ITERS); best = min(best, (time.perf_counter() - t0) * 1e3)
print({"ranks": 1, "rows": 8000, "whole_image_ms_per_step": round(best, 3)}, flush=True)
comm.close()
