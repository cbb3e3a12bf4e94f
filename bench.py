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

N > 1: BASELINE.json configs[4] -- ONE 8000 x 16000 x 3 image (rows x columns x
channels; "16000x8000x3") cut into N row slabs, strong scaling.  Every rank
owns 8000 / N rows and exchanges halo rows with its two slab neighbours every
``--halo-iters`` iterations (default 20) by grouped RCCL send/recv issued from
inside libdarsia_hip.so on a comm stream, overlapped with the interior rows
(include/darsia_hip.h: dxh_tvd_run_slab).  ``--scaling weak`` gives every rank
a 4000-row slab of a (4000 N) x 8000 x 3 image instead.  No PyTorch in this
process: torch.distributed.run is only the launcher that sets RANK / LOCAL_RANK
/ WORLD_SIZE / MASTER_PORT; the barrier and the max-over-ranks of the timing
are RCCL all-reduces through the C ABI.

Two dtype flows exist for a float32 image and BOTH are timed with the same
self-check: ``value`` is the flow of ``--promotion`` (default "native": float32 throughout,
BASELINE's float32, the reference's arithmetic under NumPy < 2); the other one -- by default
"numpy2", the product's default for float32 images, the reference's arithmetic under NumPy >= 2
-- is reported under ``other_flow`` with its own roofline block.

Rank 0 prints ONE JSON line.  ``roofline`` is for the fused sweep kernel
(k_fused_multi): algorithmic bytes per launch / mean launch duration from HIP events
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
SURVEY_VALUES_PER_ELEM = 11    # the reference's (u,f,d0,d1,b0,b1) state
# Algorithmic bytes per pixel-channel and LAUNCH of the fused sweep, per dtype flow (DESIGN.md sections 2 and 4): the launch
# reads omega*f, u, q0, q1 and writes u, q0, q1 -- 7 values, whatever the number K of outer iterations it fuses in registers.
FLOW = {
    # every quantity float32: the reference's own arithmetic under NumPy < 2 (tests/golden/tvd_np1_golden.npz), BASELINE's "float32"
    "native": {"bytes_per_elem": 7 * 4, "dtype": "f32", "storage": "7 float32 planes",
               "kernel": "k_fused_multi<float,C=%d,coupled,aniso,K=%d>", "kernel1": "k_fused<float,C=%d,coupled,aniso,scalar>"},
    # the reference's flow under NumPy >= 2 (the product's DEFAULT for float32 images): u, q float64; d, b, rhs float32 values
    # omega*img is read as a float32 plane (4 B), u / q0 / q1 are float64 planes read and written (6 x 8 B): 52 B
    "numpy2": {"bytes_per_elem": 6 * 8 + 4, "dtype": "f32 image, f64 iterate (mixed, as the reference under NumPy >= 2)",
               "storage": "6 float64 planes + omega*img as a float32 plane",
               "kernel": "k_fused_multi_np2<C=%d,coupled,K=%d>", "kernel1": "k_fused_np2<C=%d,coupled,scalar>"},
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--iters", type=int, default=100, help="TV iterations per step")
    p.add_argument("--height", type=int, default=None, help="rows (default: 4000 at N = 1 and weak scaling, 8000 for "
                                                            "the strong-scaling image of configs[4])")
    p.add_argument("--width", type=int, default=None, help="columns (default 8000; 16000 for configs[4])")
    p.add_argument("--channels", type=int, default=3)
    p.add_argument("--halo-iters", type=int, default=20,
                   help="iterations between halo exchanges (N > 1); 20 halo rows per side cost 1 %% redundant work on a "
                        "4000-row slab and leave 5 exchanges per 100-iteration step")
    p.add_argument("--scaling", choices=("weak", "strong"), default=None,
                   help="N > 1: strong (default, configs[4]) or weak")
    p.add_argument("--promotion", choices=("native", "numpy2"), default="native",
                   help="the dtype flow `value` is timed in: native = float32 throughout (BASELINE's float32; the reference "
                        "under NumPy < 2), numpy2 = the product's default for float32 images (the reference under NumPy >= 2: "
                        "float64 iterate).  The other flow is timed as well and reported under `other_flow`")
    p.add_argument("--one-flow", action="store_true", help="time only the flow of --promotion")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-self-check", action="store_true", help="skip the replays of the timed iterations")
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
        "sample": f"{h}x{w}x{channels} float32, {n} iterations, NumPy oracle (oracle/tvd_numpy.py) in the reference's dtype "
                  f"flow under the NumPy it runs with here ({np.__version__}: float64 iterate for a float32 image), {dt:.1f} s; "
                  f"host has {os.cpu_count()} logical CPUs, the reference uses one",
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


def roofline_block(flow: str, elems: float, channels: int, kernel_ms: float, launches: int, iters_total: int, traffic):
    """Roofline of the dominant kernel of one flow: algorithmic bytes per launch / mean launch duration from the HIP events
    the library records around the launches on their own stream (dxh_tvd_result.loop_ms)."""
    f = FLOW[flow]
    ms_launch = kernel_ms / launches
    k_fuse = iters_total / launches   # outer iterations per launch (temporal blocking)
    per_launch = f["bytes_per_elem"] * elems
    achieved = per_launch / (ms_launch * 1e-3) / 1e9
    kname = (f["kernel"] % (channels, round(k_fuse))) if k_fuse > 1.5 else (f["kernel1"] % channels)
    t = (traffic or {}).get(flow) if traffic and "native" in traffic else (traffic if flow == "native" else None)
    blk = {
        "bound": "hbm",
        "kernel": kname,
        "iterations_per_launch": round(k_fuse, 3),
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": t.get("hbm_bytes_per_launch") if t else None,
        "algorithmic_bytes_per_launch": per_launch,
        "ms_per_launch": round(ms_launch, 5),
        "launches_timed": launches,
        "note": f"algorithmic bytes of ONE LAUNCH of the K-stage kernel: {f['storage']} (read omega*f,u,q0,q1; write u,q0,q1) = "
                f"{f['bytes_per_elem']} B per pixel-channel, whatever the number K of outer iterations the launch fuses in "
                "registers. SURVEY.md 8(d) prices the reference's own float32 state at 11 values = 44 B per pixel-channel per "
                "ITERATION; against that accounting see survey_44B_per_iteration",
        "survey_44B_per_iteration": {
            "achieved": round(SURVEY_VALUES_PER_ELEM * 4 * elems * k_fuse / (ms_launch * 1e-3) / 1e9, 1),
            "frac": round(SURVEY_VALUES_PER_ELEM * 4 * elems * k_fuse / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        },
    }
    if t:
        blk["traffic_source"] = (t.get("source") or "") + " -- a PROFILE-TIME figure of the committed rocprofv3 passes, not " \
                                                          "collected by this run"
    return blk


def time_single(flow: str, img, a, device: int, mu: float):
    """N = 1: warm-up, the timed steps and (outside the timed region) the self-check of one dtype flow."""
    from darsia_amd import _lib
    from darsia_amd.restoration.split_bregman_tvd import DeviceTVD

    st = DeviceTVD(img.shape, img.dtype, dim=2, device=device, promotion=flow)
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
    # Outside the timed region: the state the timed launches left behind must be, bit for bit, what the
    # one-iteration-per-launch sweep (the kernel the suite pins against the oracle at this very size: tests/test_tvd_gpu.py
    # native, tests/test_numpy2_flow_gpu.py numpy2) produces for the same number of iterations -- a benchmark that skipped
    # work would fail here.
    self_check = None
    if not a.no_self_check:
        import hashlib

        got = st.download()
        st.close()
        os.environ["DXH_FUSE_K"] = "1"
        try:
            st = DeviceTVD(img.shape, img.dtype, dim=2, device=device, promotion=flow)
            st.set_coeffs(mu, 1.0, None, img)
            st.upload(img)
            n_total = a.iters * (a.warmup + a.steps)
            r1 = st.run(n_total)
            want = st.download()
        finally:
            del os.environ["DXH_FUSE_K"]
        self_check = {"iterations": n_total, "k1_replay_launches": int(r1.launches),
                      "bit_identical_to_k1_replay": bool(np.array_equal(got, want)),
                      "finite": bool(np.isfinite(got).all()),
                      "sha256_of_result": hashlib.sha256(got.tobytes()).hexdigest()[:16]}
        if not (self_check["bit_identical_to_k1_replay"] and self_check["finite"]):
            raise SystemExit(f"bench self-check failed ({flow} flow): timed state differs from the one-iteration-per-launch replay")
    st.close()
    return {"elapsed": elapsed, "kernel_ms": kernel_ms, "launches": launches, "self_check": self_check}


def time_slabs(flow: str, a, comm, rank: int, world: int, device: int, total_rows: int, W: int, C: int, mu: float):
    """N > 1: one dtype flow on row slabs -- warm-up, the timed steps between barriers, per-rank diagnostics, the same slab
    WITHOUT exchanges on this rank alone, and (outside the timed region) the single-GPU replay self-check."""
    from darsia_amd.parallel import SlabTVD, make_plan
    from darsia_amd.restoration.split_bregman_tvd import DeviceTVD
    from darsia_amd.synthetic import synthetic_image

    plan = make_plan(total_rows, world, rank, a.halo_iters)
    slab = synthetic_image(plan.rows, W, C, row0=plan.slab0)
    s = SlabTVD(plan, slab, mu=mu, dim=2, halo_iters=a.halo_iters, comm=comm, promotion=flow)

    def one_step():
        # enqueue only; the halos the next step starts from are refreshed behind this step's last launch
        s.run(a.iters, more_to_come=True, asynchronous=True)

    for _ in range(a.warmup):
        one_step()
    s.synchronize()
    comm.barrier()
    stats0 = s.slab_stats()           # (synchronises; the warm-up's counts are subtracted below)
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    t_enq = time.perf_counter()
    s.synchronize()
    t_own = time.perf_counter()       # this rank's own stream has drained (its last receive included)
    comm.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = comm.allreduce([elapsed], "max")[0]
    # Per-rank diagnostics, so that ONE record tells where the time of a disappointing curve went: every rank's own
    # loop time (enqueue -> its stream drained, before the barrier), and from the library's event triples around each
    # exchange the time the exchange stream idled until the interface rows were signalled and the time from there to
    # the end of the scatter (gather + RCCL send/recv, i.e. waiting for the neighbour, + scatter).
    stats1 = s.slab_stats()
    d = {k: stats1[k] - stats0[k] for k in ("launches", "signalling_launches", "exchanges", "exchanges_timed",
                                            "wait_timeouts", "wait_signal_ms", "exchange_ms")}
    loop_ms = (t_own - t0) * 1e3
    per_x = max(1, d["exchanges_timed"])
    # The same slab (halo rows included: the rows this rank's launches cover) as a stand-alone image on this GPU, no
    # communicator, no exchange: what the slab decomposition costs before any link time.  slab-alone / slab-with-exchange is
    # the efficiency of the exchange + overlap on this rank.
    with DeviceTVD(slab.shape, slab.dtype, dim=2, device=device, promotion=flow) as alone:
        alone.set_coeffs(mu, 1.0, None, slab)
        alone.upload(slab)
        alone.run(a.iters)
        ta = time.perf_counter()
        for _ in range(a.steps):
            alone.run(a.iters)
        alone_ms = (time.perf_counter() - ta) * 1e3 / a.steps
    mine = [loop_ms, (t_enq - t0) * 1e3, d["wait_signal_ms"] / per_x, d["exchange_ms"] / per_x, alone_ms]
    mx = comm.allreduce(mine, "max")
    mn = [-v for v in comm.allreduce([-v for v in mine], "max")]
    rank_diag = {
        "loop_ms_per_step": {"min": round(mn[0] / a.steps, 4), "max": round(mx[0] / a.steps, 4)},
        "host_enqueue_ms_per_step": {"min": round(mn[1] / a.steps, 4), "max": round(mx[1] / a.steps, 4)},
        "ms_per_exchange_waiting_for_interface_rows": {"min": round(mn[2], 4), "max": round(mx[2], 4)},
        "ms_per_exchange_signal_to_scatter_end": {"min": round(mn[3], 4), "max": round(mx[3], 4)},
        "no_exchange_ms_per_step_same_slab_alone": {"min": round(mn[4], 4), "max": round(mx[4], 4)},
        "exchange_efficiency_slab_alone_over_loop": round(mx[4] / (mx[0] / a.steps), 4),
        "rank0": {"launches": int(d["launches"]), "signalling_launches": int(d["signalling_launches"]),
                  "exchanges": int(d["exchanges"]), "exchanges_timed": int(d["exchanges_timed"]),
                  "wait_timeouts": int(d["wait_timeouts"]),
                  "wait_mode": "hipStreamWaitValue32" if stats1["wait_mode"] == 1 else "polling kernel"},
    }
    del slab
    # Outside the timed region: every rank replays ALL iterations on the WHOLE image alone, on its own GPU, and
    # compares its own rows of the distributed result with that replay, bit for bit -- the slab decomposition, the
    # halo exchange over RCCL and the overlap must not change a single value.
    self_check = None
    if not a.no_self_check:
        n_total = a.iters * (a.warmup + a.steps)
        got = s.owned_result()
        whole = synthetic_image(total_rows, W, C)
        st = DeviceTVD(whole.shape, whole.dtype, dim=2, device=device, promotion=flow)
        st.set_coeffs(mu, 1.0, None, whole)
        st.upload(whole)
        del whole
        st.run(n_total)
        replay = st.download()[plan.own0:plan.own1]
        st.close()
        same = bool(np.array_equal(got, replay))
        worst = 0.0 if same else float(np.max(np.abs(got.astype(np.float64) - replay)))
        n_same = int(round(comm.allreduce([1.0 if same else 0.0], "sum")[0]))
        worst = comm.allreduce([worst], "max")[0]
        self_check = {"iterations": n_total, "ranks_bit_identical_to_single_gpu_replay": n_same, "ranks": world,
                      "max_abs_difference": worst}
        if not same:
            print(f"[bench] rank {rank} ({flow} flow): rows {plan.own0}..{plan.own1} differ from the single-GPU replay "
                  f"(max abs {worst:.3e})", file=sys.stderr, flush=True)
        del got, replay
    s.close()
    return {"elapsed": elapsed, "rank_diag": rank_diag, "self_check": self_check, "rows_per_rank": plan.own1 - plan.own0}


def flow_words(flow: str) -> str:
    return ("native float32 flow (every quantity float32: the reference's own arithmetic under NumPy < 2, bit for bit -- "
            "tests/golden/tvd_np1_golden.npz)" if flow == "native" else
            "numpy2 flow (the reference's own arithmetic under NumPy >= 2, bit for bit: float64 iterate, float32 d / b / rhs -- "
            "tests/golden/tvd_golden.npz; the DEFAULT of darsia_amd.split_bregman_tvd for float32 images)")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        a.gpus = world

    # DXH_BENCH_REHEARSAL=1: rehearse the N-rank path on a box with fewer GPUs than ranks (ranks round-robin over the
    # visible devices, halo rows staged through the host and TCP sockets).  Never used by the driver.
    rehearsal = os.environ.get("DXH_BENCH_REHEARSAL", "")

    from darsia_amd import _lib
    from darsia_amd.synthetic import synthetic_image

    _lib.require_gpu()  # no CPU fallback: fail loudly
    C = a.channels
    mu = 0.1
    if a.scaling is None:
        a.scaling = "strong" if world > 1 else "weak"
    strong_c5 = world > 1 and a.scaling == "strong"
    H = a.height if a.height is not None else (8000 if strong_c5 else 4000)
    W = a.width if a.width is not None else (16000 if strong_c5 else 8000)
    comm_info = None
    main_flow = a.promotion
    other_flow = None if a.one_flow else ("numpy2" if main_flow == "native" else "native")
    other = None

    if world == 1:
        img = synthetic_image(H, W, C)
        res = time_single(main_flow, img, a, local_rank, mu)
        if other_flow:
            # Outside the timed region of `value`, with the same rigour: the same workload in the other dtype flow (a failure
            # there is reported in the line, it does not take `value` with it)
            try:
                other = time_single(other_flow, img, a, local_rank, mu)
            except (Exception, SystemExit) as e:  # noqa: BLE001 -- incl. a failed self-check of the OTHER flow
                other = {"error": f"{type(e).__name__}: {e}"}
        total_rows, rows_per_rank = H, H
    else:
        from darsia_amd.parallel import init_from_env

        base_port = int(os.environ.get("MASTER_PORT", "29500")) + 17 if rehearsal else None
        comm, rank, world, device = init_from_env(rehearsal_base_port=base_port)
        total_rows = H * world if a.scaling == "weak" else H
        res = time_slabs(main_flow, a, comm, rank, world, device, total_rows, W, C, mu)
        comm_info = comm.info()
        if other_flow:
            other = time_slabs(other_flow, a, comm, rank, world, device, total_rows, W, C, mu)
        comm.close()
        rows_per_rank = res["rows_per_rank"]
        # a mismatch is printed with the line (value null, "data" says so) so that the record shows what happened on
        # hardware this builder never sees -- and then every rank exits non-zero
    elapsed = res["elapsed"]
    self_check = res["self_check"]

    def failed(r):
        return bool(world > 1 and r and r["self_check"] and r["self_check"]["ranks_bit_identical_to_single_gpu_replay"] != world)

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
            "dtype": "f32" if main_flow == "native" else "f32/f64 mixed (f64 iterate)",
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: ranks share GPUs, host-staged exchange; "
                    "not a valid multi-GPU number)",
            "config": {
                "workload": ("BASELINE configs[4]: " if (total_rows, W, C) == (8000, 16000, 3) else
                             "BASELINE configs[1]: " if (total_rows, W, C) == (4000, 8000, 3) else "") +
                            f"anisotropic split-Bregman TVD, {total_rows}x{W}x{C} float32 (H x W x C), "
                            f"{a.iters} iterations per step, mu=0.1 omega=1 ell=2mu Jacobi(maxiter=1) eps=off, "
                            "channel-coupled as in the reference; `value` is the " + flow_words(main_flow) +
                            ("; the product's default for a float32 image is the OTHER flow (numpy2), timed with the same "
                             "self-check under `other_flow`" if main_flow == "native" else
                             "; BASELINE's float32 arithmetic is the OTHER flow (native), under `other_flow`"),
                "dtype_flow": main_flow,
                "iters_per_step": a.iters,
                "rows_per_gpu": rows_per_rank,
                "parallelism": "single GPU" if world == 1 else f"row slabs x{world}, halo exchange every "
                               f"{a.halo_iters} iterations: grouped RCCL send/recv from libdarsia_hip.so, overlapped "
                               "with the interior rows",
            },
        }
        if comm_info:
            out["comm"] = {"transport": comm_info["kind"], "exchanges": comm_info["exchanges"],
                           "bytes_sent_rank0": comm_info["bytes_sent"]}
            out["per_rank"] = res["rank_diag"]
        if world == 1:
            elems = float(H) * W * C
            traffic = load_traffic()
            out["roofline"] = roofline_block(main_flow, elems, C, res["kernel_ms"], res["launches"], a.iters * a.steps, traffic)
            if self_check:
                out["self_check"] = self_check
            if other and "error" in other:
                out["other_flow"] = {"promotion": other_flow, "value": None, "error": other["error"]}
            elif other:
                oe = other["elapsed"]
                out["other_flow"] = {
                    "promotion": other_flow, "what": flow_words(other_flow),
                    "value": round(pix_iters / oe / 1e6, 1), "unit": "MPix/s", "ms_per_step": round(oe / a.steps * 1e3, 4),
                    "steps": a.steps, "warmup": a.warmup,
                    "roofline": roofline_block(other_flow, elems, C, other["kernel_ms"], other["launches"], a.iters * a.steps, traffic),
                    "self_check": other["self_check"],
                }
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(C, a.cpu_sample)
        else:
            if self_check:
                out["self_check"] = self_check
            if failed(res):
                out["data"] += " -- SELF-CHECK FAILED: the distributed result differs from the single-GPU replay, this number is not valid"
                out["value"] = None      # the diagnostics stay in the record, the number does not
            if other:
                oe = other["elapsed"]
                out["other_flow"] = {
                    "promotion": other_flow, "what": flow_words(other_flow),
                    "value": None if failed(other) else round(pix_iters / oe / 1e6, 1), "unit": "MPix/s",
                    "ms_per_step": round(oe / a.steps * 1e3, 4), "per_rank": other["rank_diag"], "self_check": other["self_check"],
                }
        print(json.dumps(out), flush=True)
    if failed(res) or failed(other):
        raise SystemExit(3)              # every rank: an invalid multi-GPU result must not look like a success


if __name__ == "__main__":
    main()
