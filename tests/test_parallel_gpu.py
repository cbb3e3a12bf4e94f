"""The row-slab driver with the HIP engine: two processes share GPU 0 and
exchange halos through gloo (host-staged); the result must equal the
single-context HIP result bit for bit.  (RCCL needs one GPU per rank, which the
1-GPU test box cannot offer; the RCCL transport differs only in the backend
string.)"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from darsia_amd.parallel import SlabTVD, make_plan

        img = np.load(os.path.join(out_dir, "img.npy"))
        plan = make_plan(img.shape[0], world, rank, case["halo_iters"])
        slab = np.ascontiguousarray(img[plan.slab0:plan.slab1])
        mu = 0.1
        if case.get("het"):
            mu = np.ascontiguousarray(np.load(os.path.join(out_dir, "mu.npy"))[plan.slab0:plan.slab1])
        s = SlabTVD(plan, slab, mu=mu, dim=2, halo_iters=case["halo_iters"], isotropic=case.get("iso", False),
                    device=0)
        done = s.run(case["n_iter"], eps=case.get("eps"))
        np.save(os.path.join(out_dir, f"out_{rank}.npy"), s.owned_result())
        np.save(os.path.join(out_dir, f"done_{rank}.npy"), np.int64(done))
        s.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", [
    dict(shape=(300, 530, 3), world=2, halo_iters=4, n_iter=10),
    dict(shape=(400, 300, 3), world=2, halo_iters=20, n_iter=44),   # bench.py's exchange interval: 5 K=4 launches per round
    dict(shape=(97, 260), world=3, halo_iters=1, n_iter=5),
    dict(shape=(120, 300, 3), world=2, halo_iters=3, n_iter=80, eps=2e-3),
    dict(shape=(90, 270, 3), world=2, halo_iters=2, n_iter=6, iso=True),
    dict(shape=(210, 300, 3), world=2, halo_iters=3, n_iter=11, het=True),
], ids=["rgb_k4", "rgb_k20_bench_interval", "gray_k1_3ranks", "eps", "iso", "het_mu_array"])
def test_two_ranks_one_gpu(tmp_path, case):
    import torch.multiprocessing as mp

    import darsia_amd as da

    rng = np.random.default_rng(3)
    img = rng.random(case["shape"]).astype(np.float32)
    np.save(tmp_path / "img.npy", img)
    mu = (0.03 + 0.1 * rng.random(case["shape"])).astype(np.float32) if case.get("het") else 0.1
    if case.get("het"):
        np.save(tmp_path / "mu.npy", mu)
    world = case["world"]
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"out_{r}.npy") for r in range(world)], axis=0)
    with da.DeviceTVD(img.shape, img.dtype, dim=2) as st:
        st.set_coeffs(mu, 1.0, None, img)
        st.upload(img)
        res = st.run(case["n_iter"], eps=case.get("eps"), isotropic=case.get("iso", False))
        ref = st.download()
    dones = {int(np.load(tmp_path / f"done_{r}.npy")) for r in range(world)}
    assert dones == {res.iters_done}
    assert np.array_equal(got, ref), f"max abs {np.abs(got - ref).max():.3e}"
