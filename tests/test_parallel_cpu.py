"""world_size-2 (and 3) runs of the row-slab driver with the CPU oracle engine, over gloo and over the product's
torch-free socket transport: the N-rank result must equal the single-image result bit for bit."""
import os
import socket
import sys

import multiprocessing as mp

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from darsia_amd.parallel import SlabTVD, SocketTransport, make_plan
    from slab_oracle_engine import GlooTransport, OracleSlabEngine

    use_gloo = case.get("transport", "gloo") == "gloo"
    if use_gloo:
        # torch lives in the worker processes only: importing it into the pytest process would put a second ROCm stack
        # (PyTorch's bundled one) next to the one libdarsia_hip.so uses
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
        transport = GlooTransport()
    else:
        transport = SocketTransport(rank, world, port)
    try:

        img = np.load(os.path.join(out_dir, "img.npy"))
        plan = make_plan(img.shape[0], world, rank, case["halo_iters"], case.get("jm", 1))
        slab = np.ascontiguousarray(img[plan.slab0:plan.slab1])
        s = SlabTVD(plan, slab, mu=case["mu"], omega=case.get("omega", 1.0), ell=case.get("ell"), dim=2,
                    halo_iters=case["halo_iters"], jacobi_maxiter=case.get("jm", 1),
                    isotropic=case.get("iso", False), engine_factory=OracleSlabEngine, transport=transport)
        done = s.run(case["n_iter"], eps=case.get("eps"))
        np.save(os.path.join(out_dir, f"out_{rank}.npy"), s.owned_result())
        np.save(os.path.join(out_dir, f"done_{rank}.npy"), np.int64(done))
    finally:
        if use_gloo:
            dist.destroy_process_group()
        else:
            transport.close()


CASES = [
    dict(shape=(41, 23, 3), world=2, halo_iters=1, mu=0.1, n_iter=7),
    dict(shape=(41, 23, 3), world=2, halo_iters=3, mu=0.1, n_iter=8),      # 8 = 3 + 3 + 2
    dict(shape=(50, 17), world=3, halo_iters=2, mu=0.15, omega=0.8, n_iter=6),
    dict(shape=(40, 19, 3), world=2, halo_iters=2, mu=0.1, n_iter=5, jm=2),
    dict(shape=(36, 21), world=2, halo_iters=4, mu=0.1, n_iter=9, iso=True, dtype="float64"),
    dict(shape=(44, 20, 3), world=2, halo_iters=3, mu=0.1, n_iter=60, eps=3e-3),
    dict(shape=(50, 17), world=3, halo_iters=2, mu=0.15, omega=0.8, n_iter=7, transport="socket"),
    dict(shape=(44, 20, 3), world=3, halo_iters=3, mu=0.1, n_iter=60, eps=3e-3, transport="socket"),
]


@pytest.mark.parametrize("case", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_slabs_equal_single_image(tmp_path, case):
    from oracle import tvd_numpy as O

    rng = np.random.default_rng(42)
    img = rng.random(case["shape"]).astype(case.get("dtype", "float32"))
    np.save(tmp_path / "img.npy", img)
    world = case["world"]
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = np.concatenate([np.load(tmp_path / f"out_{r}.npy") for r in range(world)], axis=0)
    ref, _, _, done = O.split_bregman_tvd(img, mu=case["mu"], omega=case.get("omega", 1.0), max_num_iter=case["n_iter"],
                                          eps=case.get("eps"), isotropic=case.get("iso", False),
                                          jacobi_maxiter=case.get("jm", 1), promotion="strict", return_state=True)
    dones = {int(np.load(tmp_path / f"done_{r}.npy")) for r in range(world)}
    assert dones == {done}
    if case.get("eps"):
        assert 1 < done < case["n_iter"]
    assert got.shape == ref.shape and got.dtype == ref.dtype
    assert np.array_equal(got, ref), f"max abs {np.abs(got - ref).max():.3e}"


def test_plan_rejects_too_small_slabs():
    from darsia_amd.parallel import make_plan

    with pytest.raises(ValueError, match="smaller than the halo"):
        make_plan(10, 4, 0, halo_iters=3)
    p = make_plan(4000 * 8, 8, 3, halo_iters=4)
    assert (p.own1 - p.own0, p.top, p.bot) == (4000, 4, 5)
