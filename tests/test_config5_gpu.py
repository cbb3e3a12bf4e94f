"""BASELINE.json configs[4] at its own size on ONE GPU (`pytest -m gpu`): the 8000 x 16000 x 3 float32 image that
``bench.py --gpus N`` cuts into N row slabs.

(a) the whole image in one context, 20 iterations: K = 4 launches == one iteration per launch, and the result equals the
    oracle's on windows (dependence cone: jacobi.py:43-54, split_bregman_tvd.py:84-89,144 -- after n iterations a pixel
    knows nothing beyond n + 1 pixels, tests/test_tvd_gpu.py states the argument);
(b) the slab ONE rank owns at N = 8 (1000 owned rows + 20 / 21 halo rows, 16000 x 3 wide) through the product's RCCL
    path: a single rank whose slab has halos on both sides is its own neighbour, i.e. the 1000-row image is periodic
    along axis 0; 100 iterations with an exchange every 20 (bench.py's interval), interface-first launch order, the
    device-counter-triggered overlap -- bit for bit against the same rows tiled three times in one context;
(c) two host-staged ranks (RCCL refuses two ranks on one device), each with its 4000-row slab of the full image,
    against one context holding the whole image: every owned row bit-identical (compared through SHA-256 of the rows
    and a sample of them).
"""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
from oracle import tvd_numpy as O

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

H5, W5, C5 = 8000, 16000, 3
MU = 0.1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


FLOWS = ["native", "numpy2"]   # the headline's float32 flow and the product's default (the reference under NumPy >= 2)
ORACLE_FLOW = {"native": "strict", "numpy2": "numpy2"}


def _run_whole(img, n_iter, promotion="native"):
    with da.DeviceTVD(img.shape, img.dtype, dim=2, promotion=promotion) as st:
        st.set_coeffs(MU, 1.0, None, img)
        st.upload(img)
        res = st.run(n_iter)
        return st.download(), res


@pytest.mark.parametrize("flow", FLOWS)
def test_config5_whole_image_on_one_gpu_windows_against_the_oracle(monkeypatch, flow):
    img = synthetic_image(H5, W5, C5)
    n = 20
    out, res = _run_whole(img, n, flow)
    assert res.iters_done == n and res.launches == n // 4
    monkeypatch.setenv("DXH_FUSE_K", "1")
    out1, res1 = _run_whole(img, n, flow)
    monkeypatch.delenv("DXH_FUSE_K")
    assert res1.launches == n
    assert np.array_equal(out, out1)
    del out1
    assert np.isfinite(out).all()
    m, S = n + 2, 256
    for (r0, c0) in ((0, 0), ((H5 - S) // 2, (W5 - S) // 2), (H5 - S, W5 - S), (0, W5 - S), (H5 - S, 0),
                     (3999 - S // 2, 7 * 208 * 4 // 3)):   # the last one straddles the two-rank slab interface
        crop = np.ascontiguousarray(img[r0:r0 + S, c0:c0 + S])
        ra, rb = (0 if r0 == 0 else m), (S if r0 + S == H5 else S - m)
        ca, cb = (0 if c0 == 0 else m), (S if c0 + S == W5 else S - m)
        got = out[r0 + ra:r0 + rb, c0 + ca:c0 + cb]
        want = O.split_bregman_tvd(crop, mu=MU, max_num_iter=n, promotion=ORACLE_FLOW[flow])[ra:rb, ca:cb]
        assert np.array_equal(got, want), f"window ({r0},{c0}): max abs {np.abs(got - want).max():.3e}"
        want2 = O.split_bregman_tvd(crop, mu=MU, max_num_iter=n)[ra:rb, ca:cb]      # the reference's NumPy-2 dtype flow
        np.testing.assert_allclose(got, want2, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("flow", FLOWS)
def test_config5_slab_of_one_rank_at_n8_over_rccl(flow):
    """1000 owned rows x 16000 x 3 with 20 + 21 halo rows: what rank r of 8 holds; neighbour = itself (periodic)."""
    from darsia_amd.parallel import Comm, SlabTVD, make_plan, periodic_slab

    rows, n, k = H5 // 8, 100, 20
    img = synthetic_image(rows, W5, C5, row0=3 * rows)      # the rows rank 3 of 8 owns
    os.environ["MASTER_PORT"] = str(_free_port())
    comm = Comm.rccl(0, 1, device=0)
    try:
        plan = make_plan(rows, 1, 0, k, periodic=True)
        assert plan.rows == rows + 2 * k + 1
        s = SlabTVD(plan, periodic_slab(img, plan), mu=MU, halo_iters=k, comm=comm, promotion=flow)
        done = 0
        for part in (60, 40):       # two calls continue one sequence, the first prefetches the halos for the second
            done += s.run(part, more_to_come=done + part < n, asynchronous=True)
        got = s.owned_result()
        info = comm.info()
        stats = s.slab_stats()
        s.close()
    finally:
        comm.close()
    assert done == n and info["exchanges"] == n // k - 1
    # every exchange was started by the device counter behind a signalling launch, none timed out
    assert stats["signalling_launches"] == info["exchanges"] and stats["wait_timeouts"] == 0
    tiled = np.ascontiguousarray(np.concatenate([img, img, img], axis=0))
    ref, _ = _run_whole(tiled, n, flow)
    ref = ref[rows:2 * rows]
    assert np.array_equal(got, ref), f"max abs {np.abs(got - ref).max():.3e}"


def _rank_worker(rank, world, port, n_iter, halo_iters, out_dir, flow):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    from darsia_amd.parallel import Comm, SlabTVD, SocketTransport, make_plan

    comm = Comm.host(SocketTransport(rank, world, port), device=0)
    try:
        plan = make_plan(H5, world, rank, halo_iters)
        slab = synthetic_image(plan.rows, W5, C5, row0=plan.slab0)
        s = SlabTVD(plan, slab, mu=MU, halo_iters=halo_iters, comm=comm, promotion=flow)
        del slab
        done = s.run(n_iter)
        mine = s.owned_result()
        s.close()
        with open(os.path.join(out_dir, f"sha_{rank}.txt"), "w") as f:
            f.write(f"{done} {hashlib.sha256(mine.tobytes()).hexdigest()}")
        np.save(os.path.join(out_dir, f"sample_{rank}.npy"), mine[::499, ::53])
    finally:
        comm.close()


@pytest.mark.parametrize("flow", FLOWS)
def test_config5_two_host_staged_ranks_equal_one_context(tmp_path, flow):
    import multiprocessing as mp

    n, k, world = 44, 20, 2     # two exchanges, the last round shorter than the halo budget
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, n, k, str(tmp_path), flow)) for r in range(world)]
    for p in procs:
        p.start()
    whole = synthetic_image(H5, W5, C5)
    ref, _ = _run_whole(whole, n, flow)
    del whole
    for p in procs:
        p.join(900)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for r in range(world):
        o0, o1 = r * H5 // world, (r + 1) * H5 // world
        done, sha = open(tmp_path / f"sha_{r}.txt").read().split()
        assert int(done) == n
        sample = np.load(tmp_path / f"sample_{r}.npy")
        assert np.array_equal(sample, ref[o0:o1][::499, ::53]), f"rank {r}: sampled rows differ"
        assert sha == hashlib.sha256(np.ascontiguousarray(ref[o0:o1]).tobytes()).hexdigest(), f"rank {r}"
