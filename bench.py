#!/usr/bin/env python3
"""Headline benchmark: MPix/s per TV iteration of the anisotropic split-Bregman
TVD sweep on a 4000 x 8000 x 3 float32 synthetic image (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A STEP is one pass of the hot path: ``--iters`` (100) outer iterations of
``split_bregman_tvd`` (mu = 0.1, omega = 1, ell = 2 mu, Jacobi(maxiter=1), eps
off) over the device-resident image.  Steps continue the same iteration
sequence (no re-upload inside the timed region).

N > 1: weak scaling.  Every rank owns a 4000-row slab of ONE image of 4000*N
rows (N = 4: the 16000 x 8000 x 3 image of configs[4]) and exchanges halo rows
with its two neighbours every ``--halo-iters`` iterations (default 20) over RCCL
(darsia_amd/parallel.py).  ``--scaling strong`` splits the 4000-row image
instead.

Rank 0 prints ONE JSON line.  ``roofline`` is for the fused sweep kernel
(k_fused): algorithmic bytes per launch / mean launch duration from HIP events
on the kernel's own stream.  ``cpu_baseline`` times the
NumPy oracle (the reference's execution model: one thread) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_VALUE = 4            # float32
VALUES_PER_ELEM = 7            # k_fused: read of,u,q0,q1 ; write u,q0,q1  
SURVEY_VALUES_PER_ELEM = 11    # the reference's (u,f,d0,d1,b0,b1) state


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--iters", type=int, default=100, help="TV iterations per step")
    p.add_argument("--height", type=int, default=4000)
    p.add_argument("--width", type=int, default=8000)
    p.add_argument("--channels", type=int, default=3)
    p.add_argument("--halo-iters", type=int, default=20,
                   help="iterations between halo exchanges (N > 1); 20 halo rows per side cost 1 %% redundant work on a "
                        "4000-row slab and leave 5 exchanges per 100-iteration step")
    p.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample", default="4000x8000", help="HxW of the CPU baseline sample")
    return p.parse_args()


def cpu_baseline(channels: int, sample: str):
    """NumPy oracle (reference dtype flow, 1 thread) on a bounded sample of the same workload."""
    from darsia_amd.synthetic import synthetic_image
    from oracle import tvd_numpy as oracle

    h, w = (int(v) for v in sample.split("x"))
    img = synthetic_image(h, w, channels)
    oracle.split_bregman_tvd(img, mu=0.1, max_num_iter=1)  # warm-up (page faults, imports)
    n = 3
    t0 = time.perf_counter()
    oracle.split_bregman_tvd(img, mu=0.1, max_num_iter=n)
    dt = time.perf_counter() - t0
    return {
        "value": round(h * w * n / dt / 1e6, 3),
        "unit": "MPix/s per TV iteration",
        "cores": 1,
        "kind": "port",
        "sample": f"{h}x{w}x{channels} float32, {n} iterations, NumPy oracle (oracle/tvd_numpy.py, reference "
                  f"dtype flow), {dt:.1f} s; host has {os.cpu_count()} logical CPUs, the reference uses one",
    }


def load_traffic():
    """HBM bytes per k_fused launch from the committed rocprofv3 PMC passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t
    except Exception:
        return None


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        a.gpus = world

    rehearsal = os.environ.get("DXH_BENCH_REHEARSAL", "")
    if world > 1:
        # torch's bundled HIP runtime must initialise before libdarsia_hip.so pulls in the system one
        import torch

        if rehearsal:
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        torch.zeros(1, device="cuda")

    from darsia_amd import _lib
    from darsia_amd.synthetic import synthetic_image

    _lib.require_gpu()  # no CPU fallback: fail loudly
    C = a.channels
    H, W = a.height, a.width
    mu = 0.1

    if world == 1:
        from darsia_amd.restoration.split_bregman_tvd import DeviceTVD

        img = synthetic_image(H, W, C)
        st = DeviceTVD(img.shape, img.dtype, dim=2, device=local_rank)
        st.set_coeffs(mu, 1.0, None, img)
        st.upload(img)
        for _ in range(a.warmup):
            st.run(a.iters)
        _lib.check(st.lib.dxh_tvd_sync(st._h))
        kernel_ms, launches = 0.0, 0
        t0 = time.perf_counter()
        for _ in range(a.steps):
            r = st.run(a.iters)      # returns after the stream has drained
            kernel_ms += r.loop_ms
            launches += r.launches
        _lib.check(st.lib.dxh_tvd_sync(st._h))
        elapsed = time.perf_counter() - t0
        total_rows, rows_per_rank = H, H
        st.close()
    else:
        import torch
        import torch.distributed as dist

        from darsia_amd.parallel import SlabTVD, make_plan

        # DXH_BENCH_REHEARSAL=gloo: rehearse the N-rank path on a box with fewer GPUs than ranks (all ranks on
        # the visible devices round-robin, halo exchange staged through the host).  Never used by the driver.
        if rehearsal:
            dist.init_process_group(rehearsal)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        total_rows = H * world if a.scaling == "weak" else H
        plan = make_plan(total_rows, world, rank, a.halo_iters)
        slab = synthetic_image(plan.rows, W, C, row0=plan.slab0)
        s = SlabTVD(plan, slab, mu=mu, dim=2, halo_iters=a.halo_iters, device=local_rank)
        rows_per_rank = plan.own1 - plan.own0

        def one_step():
            # a step ends with an exchange so that the next step starts from valid halos
            s.run(a.iters)
            s.exchange()

        for _ in range(a.warmup):
            one_step()
        s.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            one_step()
        s.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        kernel_ms, launches = None, None
        s.close()

    if rank == 0:
        pix_iters = float(total_rows) * W * a.iters * a.steps
        value = pix_iters / elapsed / 1e6
        out = {
            "metric": "MPix/s per TV iteration (anisotropic split-Bregman TVD, 8000x4000x3 float32)",
            "value": round(value, 1),
            "unit": "MPix/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not os.environ.get("DXH_BENCH_REHEARSAL") else "synthetic (REHEARSAL: ranks share GPUs, "
                    "host-staged exchange; not a valid multi-GPU number)",
            "config": {
                "workload": f"anisotropic split-Bregman TVD, {total_rows}x{W}x{C} float32 (H x W x C), "
                            f"{a.iters} iterations per step, mu=0.1 omega=1 ell=2mu Jacobi(maxiter=1) eps=off, "
                            "channel-coupled as in the reference",
                "iters_per_step": a.iters,
                "rows_per_gpu": rows_per_rank,
                "parallelism": "single GPU" if world == 1 else f"row slabs x{world}, halo exchange every "
                               f"{a.halo_iters} iterations over RCCL send/recv",
            },
        }
        if world == 1:
            elems = float(H) * W * C
            ms_launch = kernel_ms / launches
            k_fuse = a.iters * a.steps / launches   # outer iterations per launch (temporal blocking)
            achieved = VALUES_PER_ELEM * BYTES_PER_VALUE * elems / (ms_launch * 1e-3) / 1e9
            traffic = load_traffic()
            kname = ("k_fused_multi<float,C=3,coupled,aniso,K=%d>" % round(k_fuse)) if k_fuse > 1.5 else \
                "k_fused<float,C=3,coupled,aniso,scalar>"
            out["roofline"] = {
                "bound": "hbm",
                "kernel": kname,
                "iterations_per_launch": round(k_fuse, 3),
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic.get("hbm_bytes_per_launch") if traffic else None,
                "algorithmic_bytes_per_launch": VALUES_PER_ELEM * BYTES_PER_VALUE * elems,
                "ms_per_launch": round(ms_launch, 5),
                "launches_timed": launches,
                "note": "algorithmic bytes of ONE LAUNCH: 7 values (read omega*f,u,q0,q1; write u,q0,q1) x 4 B per "
                        "pixel-channel, whatever the number K of outer iterations the launch fuses in registers. "
                        "SURVEY.md 8(d) prices the reference's own state at 11 values = 44 B per pixel-channel per "
                        "ITERATION; against that accounting see survey_44B_per_iteration",
                "survey_44B_per_iteration": {
                    "achieved": round(SURVEY_VALUES_PER_ELEM * BYTES_PER_VALUE * elems * k_fuse
                                      / (ms_launch * 1e-3) / 1e9, 1),
                    "frac": round(SURVEY_VALUES_PER_ELEM * BYTES_PER_VALUE * elems * k_fuse
                                  / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                },
            }
            if traffic:
                out["roofline"]["traffic_source"] = traffic.get("source")
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(C, a.cpu_sample)
        print(json.dumps(out), flush=True)

    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
