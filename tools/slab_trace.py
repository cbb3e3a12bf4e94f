#!/usr/bin/env python3
"""One periodic-slab step (rank of configs[4] at N = ROWS_DIV) for a rocprofv3 --kernel-trace timeline."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from darsia_amd import _lib
from darsia_amd.parallel import Comm, SlabTVD, make_plan, periodic_slab
from darsia_amd.synthetic import synthetic_image
n = int(os.environ.get("ROWS_DIV", "8"))
mode = os.environ.get("MODE", "overlap")
k = int(os.environ.get("HALO", "20"))
os.environ.setdefault("MASTER_PORT", "29613")
comm = Comm.rccl(0, 1, device=0)
img = synthetic_image(8000 // n, 16000, 3)
plan = make_plan(img.shape[0], 1, 0, k, periodic=True)
s = SlabTVD(plan, periodic_slab(img, plan), mu=0.1, halo_iters=k, comm=comm, promotion='native',
            flags=_lib.FLAG_NO_OVERLAP if mode == "serial" else 0)
s.run(100, more_to_come=True, asynchronous=True)
s.synchronize()
t0 = time.perf_counter()
t_host = 0.0
for _ in range(2):
    h0 = time.perf_counter()
    s.run(100, more_to_come=True, asynchronous=True)
    t_host += time.perf_counter() - h0
s.synchronize()
print(f"mode={mode} ms_per_step={(time.perf_counter() - t0) / 2 * 1e3:.3f} host_enqueue_ms_per_step={t_host / 2 * 1e3:.3f}")
s.close(); comm.close()
