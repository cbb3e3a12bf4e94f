"""The row-slab driver with the HIP engine.

* 2-3 processes share GPU 0 and exchange halos through the library's host-staged transport (RCCL refuses several
  ranks on one device): the result must equal the single-context HIP result bit for bit, with and without the
  interface-first / overlapped launch order of ``dxh_tvd_run_slab``.
* RCCL itself: a one-rank communicator (all-reduce, barrier, slab run) on the 1-GPU box, and -- when the box has at
  least two GPUs -- one rank per GPU over RCCL send/recv, again bit-exact against one context.
"""
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
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    from darsia_amd import _lib
    from darsia_amd.parallel import Comm, SlabTVD, SocketTransport, make_plan

    if case.get("rccl"):
        comm = Comm.rccl(rank, world, device=rank)
    else:
        comm = Comm.host(SocketTransport(rank, world, port), device=0)
    try:
        img = np.load(os.path.join(out_dir, "img.npy"))
        jm = case.get("jm", 1)
        plan = make_plan(img.shape[0], world, rank, case["halo_iters"], jm)
        slab = np.ascontiguousarray(img[plan.slab0:plan.slab1])
        mu = 0.1
        if case.get("het"):
            mu = np.ascontiguousarray(np.load(os.path.join(out_dir, "mu.npy"))[plan.slab0:plan.slab1])
        flags = _lib.FLAG_NO_OVERLAP if case.get("no_overlap") else 0
        s = SlabTVD(plan, slab, mu=mu, ell=case.get("ell"), dim=case.get("dim", 2), halo_iters=case["halo_iters"], jacobi_maxiter=jm,
                    isotropic=case.get("iso", False), comm=comm, flags=flags, promotion=case.get("promotion", "native"))
        done = 0
        for part in case.get("parts", [case["n_iter"]]):
            # several calls continue one iteration sequence; all but the last prefetch the halos for the next
            done += s.run(part, eps=case.get("eps"), more_to_come=(done + part < case["n_iter"]),
                          asynchronous=case.get("async", False))
        np.save(os.path.join(out_dir, f"out_{rank}.npy"), s.owned_result())
        np.save(os.path.join(out_dir, f"done_{rank}.npy"), np.int64(done))
        np.save(os.path.join(out_dir, f"xchg_{rank}.npy"), np.int64(comm.info()["exchanges"]))
        s.close()
    finally:
        comm.close()


def _spawn(world, case, tmp_path):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def _reference(img, mu, case):
    import darsia_amd as da

    with da.DeviceTVD(img.shape, img.dtype, dim=case.get("dim", 2), promotion=case.get("promotion", "native")) as st:
        st.set_coeffs(mu, 1.0, case.get("ell"), img)
        st.upload(img)
        res = st.run(case["n_iter"], jacobi_maxiter=case.get("jm", 1), eps=case.get("eps"),
                     isotropic=case.get("iso", False))
        return st.download(), res.iters_done


CASES = {
    "rgb_k4": dict(shape=(300, 530, 3), world=2, halo_iters=4, n_iter=10),
    # bench.py's exchange interval: 5 K=4 launches per round, overlapped exchanges
    "rgb_k20_bench_interval": dict(shape=(400, 300, 3), world=2, halo_iters=20, n_iter=44),
    "rgb_k20_no_overlap": dict(shape=(400, 300, 3), world=2, halo_iters=20, n_iter=44, no_overlap=True),
    # three ranks: the middle one has two interfaces; several chunks per slab so that interface / interior parts differ
    "rgb_k8_3ranks_tall": dict(shape=(1500, 420, 3), world=3, halo_iters=8, n_iter=30),
    "rgb_k20_steps": dict(shape=(900, 300, 3), world=2, halo_iters=20, n_iter=100, parts=[40, 40, 20], **{"async": True}),
    "gray_k1_3ranks": dict(shape=(97, 260), world=3, halo_iters=1, n_iter=5),
    "gray_k6_tall": dict(shape=(2000, 520), world=2, halo_iters=6, n_iter=19),
    "eps": dict(shape=(120, 300, 3), world=2, halo_iters=3, n_iter=80, eps=2e-3),
    "iso": dict(shape=(90, 270, 3), world=2, halo_iters=2, n_iter=6, iso=True),
    "het_mu_array": dict(shape=(210, 300, 3), world=2, halo_iters=3, n_iter=11, het=True),
    # array-valued mu with a scalar ell: the per-element-threshold pipeline (k_fused_multi<..., KQV>) in slabs, with eps
    "het_mu_array_scalar_ell": dict(shape=(900, 300, 3), world=3, halo_iters=6, n_iter=25, het=True, ell=0.3),
    "het_mu_array_scalar_ell_eps": dict(shape=(260, 300, 3), world=2, halo_iters=3, n_iter=60, het=True, ell=0.3, eps=4e-3),
    "jacobi_maxiter2": dict(shape=(160, 200, 3), world=2, halo_iters=3, n_iter=7, jm=2),
    "f64": dict(shape=(600, 200, 3), world=2, halo_iters=4, n_iter=12, dtype="float64"),
    # the reference's float32 dtype flow (k_fused_np2) on slabs: scalar and float64-array mu, three ranks, eps
    "numpy2_rgb": dict(shape=(330, 300, 3), world=2, halo_iters=5, n_iter=17, promotion="numpy2", ell=1.0),
    "numpy2_gray_mu64_3ranks": dict(shape=(500, 420), world=3, halo_iters=4, n_iter=13, promotion="numpy2", het=True, mu64=True, ell=0.5),
    "numpy2_eps": dict(shape=(150, 300, 3), world=2, halo_iters=3, n_iter=60, promotion="numpy2", ell=0.2, eps=4e-3),
}


@pytest.mark.parametrize("name", list(CASES))
def test_ranks_on_one_gpu(tmp_path, name):
    case = CASES[name]
    rng = np.random.default_rng(3)
    img = rng.random(case["shape"]).astype(case.get("dtype", "float32"))
    np.save(tmp_path / "img.npy", img)
    mu = (0.03 + 0.1 * rng.random(case["shape"])).astype(np.float64 if case.get("mu64") else np.float32) if case.get("het") else 0.1
    if case.get("het"):
        np.save(tmp_path / "mu.npy", mu)
    world = case["world"]
    _spawn(world, case, tmp_path)
    got = np.concatenate([np.load(tmp_path / f"out_{r}.npy") for r in range(world)], axis=0)
    ref, ref_done = _reference(img, mu, case)
    dones = {int(np.load(tmp_path / f"done_{r}.npy")) for r in range(world)}
    assert dones == {ref_done}
    assert np.array_equal(got, ref), f"max abs {np.abs(got - ref).max():.3e}"
    if not case.get("eps"):
        # exchanges happen exactly when the halos are used up: ceil(n / k) - 1 of them
        k = case["halo_iters"]
        expect = -(-case["n_iter"] // k) - 1
        assert {int(np.load(tmp_path / f"xchg_{r}.npy")) for r in range(world)} == {expect}


def test_rccl_single_rank_communicator(tmp_path):
    """RCCL loads, a one-rank communicator initialises on the device, reduces and carries a slab run."""
    import darsia_amd as da
    from darsia_amd.parallel import Comm, SlabTVD, make_plan

    os.environ["MASTER_PORT"] = str(_free_port())
    comm = Comm.rccl(0, 1, device=0)
    try:
        assert comm.info()["kind"] == "rccl"
        assert comm.allreduce([1.5, -2.0], "sum") == [1.5, -2.0]
        assert comm.allreduce([3.0], "max") == [3.0]
        comm.barrier()
        img = np.random.default_rng(5).random((130, 300, 3)).astype(np.float32)
        plan = make_plan(img.shape[0], 1, 0, 4)
        s = SlabTVD(plan, img, mu=0.1, halo_iters=4, comm=comm)
        assert s.run(9) == 9
        got = s.owned_result()
        s.close()
    finally:
        comm.close()
    ref = da.split_bregman_tvd(img, mu=0.1, max_num_iter=9)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("order", ["lib_then_torch", "torch_then_lib"])
def test_rccl_next_to_pytorchs_rocm_stack(order):
    """A process that also imports PyTorch holds two ROCm stacks (PyTorch bundles its own HIP runtime and RCCL).  The
    communicator must take the RCCL that sits on the HIP runtime libdarsia_hip.so is bound to, whichever was loaded
    first (dxh_comm.hip: rccl::load)."""
    import subprocess

    code = (
        "import os, sys\n"
        "sys.path.insert(0, '.')\n"
        "os.environ['MASTER_PORT'] = '%d'\n"
        "order = '%s'\n"
        "if order == 'torch_then_lib':\n"
        "    import torch\n"
        "from darsia_amd import _lib\n"
        "_lib.require_gpu()\n"
        "if order == 'lib_then_torch':\n"
        "    import torch\n"
        "from darsia_amd.parallel import Comm\n"
        "c = Comm.rccl(0, 1, device=0)\n"
        "assert c.allreduce([2.0, 3.0]) == [2.0, 3.0]\n"
        "c.close()\n"
        "print('RCCL_OK')\n") % (_free_port(), order)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


@pytest.mark.parametrize("case", [
    dict(shape=(700, 420, 3), halo_iters=8, n_iter=40, flags=0),
    dict(shape=(700, 420, 3), halo_iters=8, n_iter=40, flags="no_overlap"),
    dict(shape=(1500, 260), halo_iters=20, n_iter=100, flags=0, parts=[60, 40]),
    dict(shape=(300, 200, 3), halo_iters=1, n_iter=7, flags=0),
    dict(shape=(700, 420, 3), halo_iters=8, n_iter=40, flags=0, wait="2"),
], ids=["rgb_k8", "rgb_k8_no_overlap", "gray_k20_steps", "rgb_k1", "rgb_k8_polling_kernel"])
def test_rccl_self_neighbour_periodic(case, monkeypatch):
    """RCCL send/recv on ONE GPU: a single rank whose slab has halos on both sides exchanges with itself, which makes
    the problem periodic along axis 0.  Reference: the same image tiled three times with replicate ends -- after n
    iterations the ends have influenced at most n + 1 rows, so the middle copy is the periodic result exactly.  Runs the
    real RCCL enqueue path, the comm stream, the interface-first launch order and the overlap."""
    import darsia_amd as da
    from darsia_amd import _lib
    from darsia_amd.parallel import Comm, SlabTVD, make_plan, periodic_slab

    if case.get("wait"):
        # the comm stream waits for the interface-wave counter with a polling kernel instead of hipStreamWaitValue32
        # (what a device without that stream operation gets)
        monkeypatch.setenv("DXH_SLAB_WAIT", case["wait"])
    img = np.random.default_rng(8).random(case["shape"]).astype(np.float32)
    H, n = img.shape[0], case["n_iter"]
    assert n + 2 < H
    os.environ["MASTER_PORT"] = str(_free_port())
    comm = Comm.rccl(0, 1, device=0)
    try:
        plan = make_plan(H, 1, 0, case["halo_iters"], periodic=True)
        flags = _lib.FLAG_NO_OVERLAP if case["flags"] == "no_overlap" else 0
        s = SlabTVD(plan, periodic_slab(img, plan), mu=0.1, halo_iters=case["halo_iters"], comm=comm, flags=flags)
        done = 0
        for part in case.get("parts", [n]):
            done += s.run(part, more_to_come=done + part < n, asynchronous=True)
        got = s.owned_result()
        xchg = comm.info()["exchanges"]
        s.close()
    finally:
        comm.close()
    tiled = np.ascontiguousarray(np.concatenate([img, img, img], axis=0))
    ref = da.split_bregman_tvd(tiled, mu=0.1, max_num_iter=n)[H:2 * H]
    assert np.array_equal(got, ref), f"max abs {np.abs(got - ref).max():.3e}"
    assert xchg == -(-n // case["halo_iters"]) - 1


def test_polling_wait_timeout_raises(monkeypatch):
    """The polling wait (devices without hipStreamWaitValue32) gives up after its time-out so that the stream cannot hang;
    the exchange behind it has then packed rows nobody signalled.  That must surface as an error at the next
    synchronising call (and stay one until the next upload), not as a silently wrong result.  The lost signal is
    simulated: the wait is told to expect one wave more than the launch has."""
    from darsia_amd import _lib
    from darsia_amd.parallel import Comm, SlabTVD, make_plan, periodic_slab

    monkeypatch.setenv("DXH_SLAB_WAIT", "2")
    monkeypatch.setenv("DXH_SLAB_WAIT_TIMEOUT_MS", "40")
    monkeypatch.setenv("DXH_SLAB_TEST_LOST_SIGNAL", "1")
    img = np.random.default_rng(8).random((300, 200, 3)).astype(np.float32)
    os.environ["MASTER_PORT"] = str(_free_port())
    comm = Comm.rccl(0, 1, device=0)
    try:
        plan = make_plan(img.shape[0], 1, 0, 4, periodic=True)
        s = SlabTVD(plan, periodic_slab(img, plan), mu=0.1, halo_iters=4, comm=comm)
        with pytest.raises(_lib.HipError, match="timed out"):
            s.run(12)
        with pytest.raises(_lib.HipError, match="timed out"):
            s.synchronize()
        with pytest.raises(_lib.HipError, match="timed out"):
            s.owned_result()
        st = s.slab_stats()
        assert st["wait_timeouts"] >= 1 and st["wait_mode"] == 2
        # a fresh upload starts a fresh sequence; without the lost signal it runs clean
        s.close()
        monkeypatch.delenv("DXH_SLAB_TEST_LOST_SIGNAL")
        s = SlabTVD(plan, periodic_slab(img, plan), mu=0.1, halo_iters=4, comm=comm)
        assert s.run(12) == 12
        st = s.slab_stats()
        assert st["wait_timeouts"] == 0 and st["exchanges"] == 2 and st["signalling_launches"] == 2
        assert st["exchanges_timed"] == 2 and st["exchange_ms"] > 0.0
        s.close()
    finally:
        comm.close()


def test_rccl_one_rank_per_gpu(tmp_path):
    """The product transport: one process per GPU, RCCL send/recv between slab neighbours (needs >= 2 GPUs)."""
    from darsia_amd import _lib

    ngpu = _lib.device_count()
    if ngpu < 2:
        pytest.skip("needs at least two GPUs (RCCL refuses two ranks on one device)")
    world = min(ngpu, 4)
    case = dict(shape=(1200, 700, 3), world=world, halo_iters=8, n_iter=40, rccl=True)
    img = np.random.default_rng(11).random(case["shape"]).astype(np.float32)
    np.save(tmp_path / "img.npy", img)
    _spawn(world, case, tmp_path)
    got = np.concatenate([np.load(tmp_path / f"out_{r}.npy") for r in range(world)], axis=0)
    ref, _ = _reference(img, 0.1, case)
    assert np.array_equal(got, ref), f"max abs {np.abs(got - ref).max():.3e}"
