"""bench.py contract checks on a real GPU: one JSON line with the required fields at N = 1, and the
N-rank code path in rehearsal mode (ranks share the GPU, host-staged exchange over local sockets; no torch in bench.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(stdout: str):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_single_gpu_line_has_contract_fields():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--height", "512", "--width", "768",
                        "--iters", "20", "--cpu-sample", "128x192"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1.5
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    assert "workload" in d["config"]
    assert d["config"]["dtype_flow"] == "native" and "native float32 flow" in d["config"]["workload"]
    assert "NumPy" in d["cpu_baseline"]["sample"]
    # next to `value` (the native float32 flow): the same workload in the default dtype flow of float32 images, timed over the
    # same steps, with its own roofline block and the same K = 1 replay self-check
    # (at this launch-bound test size the two flows take about the same time; at the benchmark size native is ~2 x faster)
    o = d["other_flow"]
    assert o["promotion"] == "numpy2" and 0 < o["value"] < 2 * d["value"] and o["steps"] == 2
    assert o["roofline"]["bound"] == "hbm" and 0 < o["roofline"]["frac"] < 1.5 and "np2" in o["roofline"]["kernel"]
    assert o["roofline"]["algorithmic_bytes_per_launch"] > d["roofline"]["algorithmic_bytes_per_launch"]
    assert d["self_check"]["bit_identical_to_k1_replay"] is True and o["self_check"]["bit_identical_to_k1_replay"] is True
    assert d["self_check"]["sha256_of_result"] != o["self_check"]["sha256_of_result"]


def test_single_gpu_line_in_the_default_flow():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--height", "256", "--width", "512",
                        "--iters", "12", "--promotion", "numpy2", "--one-flow", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["config"]["dtype_flow"] == "numpy2" and "other_flow" not in d and d["dtype"].startswith("f32/f64")
    assert d["self_check"]["bit_identical_to_k1_replay"] is True


def test_two_rank_rehearsal_runs_the_slab_path():
    env = dict(os.environ, DXH_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--height", "256", "--width", "512", "--iters", "16"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    # default for N > 1: strong scaling, the image's rows are split over the ranks
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["rows_per_gpu"] == 128
    assert "REHEARSAL" in d["data"] and d["value"] > 0
    assert d["comm"]["transport"] == "host" and d["comm"]["exchanges"] >= 2
    # per-rank diagnostics: loop time min / max over the ranks, exchange timings, launch and exchange counts of rank 0
    pr = d["per_rank"]
    assert 0 < pr["loop_ms_per_step"]["min"] <= pr["loop_ms_per_step"]["max"]
    assert pr["rank0"]["exchanges"] >= 2 and pr["rank0"]["exchanges_timed"] == pr["rank0"]["exchanges"] and pr["rank0"]["wait_timeouts"] == 0
    assert pr["ms_per_exchange_signal_to_scatter_end"]["max"] > 0
    # every rank compared its rows with a single-GPU replay of all iterations on the whole image
    assert d["self_check"] == {"iterations": 48, "ranks_bit_identical_to_single_gpu_replay": 2, "ranks": 2,
                               "max_abs_difference": 0.0}
    # the slab alone (no exchange) on each rank, and the same run in the product's default flow with its own diagnostics
    assert pr["no_exchange_ms_per_step_same_slab_alone"]["max"] > 0 and pr["exchange_efficiency_slab_alone_over_loop"] > 0
    o = d["other_flow"]
    assert o["promotion"] == "numpy2" and o["value"] > 0 and o["self_check"] == d["self_check"]
    assert o["per_rank"]["rank0"]["exchanges"] == pr["rank0"]["exchanges"]
    # weak scaling stays available
    cmd2 = list(cmd)
    cmd2[cmd2.index("--master-port") + 1] = str(_free_port())
    r = subprocess.run(cmd2 + ["--scaling", "weak"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["scaling"] == "weak" and d["config"]["rows_per_gpu"] == 256
