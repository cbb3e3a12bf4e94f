"""Row-slab decomposition of one TV problem over several GPUs (one process per
GPU, ``torch.distributed``; backend ``nccl`` = RCCL over xGMI on MI355X).

The reference is single-process, so nothing here mirrors
reference code; what must hold is that the N-rank result equals the 1-rank
result BIT FOR BIT (same arithmetic per pixel), which ``tests/test_parallel_cpu.py``
checks with world_size 2 on gloo.

Decomposition.  The image is cut along axis 0 into contiguous slabs.  One outer
iteration with ``m`` Jacobi sweeps reads one row above and below per sweep and
the split variables one row above; the fused kernel additionally needs the new
iterate one row below.  A slab edge that is not the image boundary therefore
loses ``m`` valid rows per iteration at the top and ``m`` (+1 once) at the
bottom.  With halos of ``k*m`` rows above and ``k*m + 1`` below, ``k`` iterations
can run between two exchanges; the halos are recomputed redundantly, which costs
(2km+1)/rows extra work and turns 3 small messages per iteration into one
larger one every ``k`` iterations (xGMI is point-to-point: each slab talks to its
two neighbours only, no ring, no collective on the data path).

A global ``eps`` stopping test needs the increment norm every iteration: then
``k = 1`` and one 8-byte all-reduce per iteration is added, as in the reference's
per-iteration test (split_bregman_tvd.py:161-175).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib


@dataclass
class SlabPlan:
    """Rows of rank ``rank``: owned [own0, own1), stored [slab0, slab1) (halos included)."""
    H: int
    world: int
    rank: int
    halo_top_rows: int
    halo_bot_rows: int

    def __post_init__(self):
        if self.world < 1 or not (0 <= self.rank < self.world):
            raise ValueError("bad rank/world")
        self.own0 = self.rank * self.H // self.world
        self.own1 = (self.rank + 1) * self.H // self.world
        self.top = 0 if self.rank == 0 else self.halo_top_rows
        self.bot = 0 if self.rank == self.world - 1 else self.halo_bot_rows
        self.slab0 = self.own0 - self.top
        self.slab1 = self.own1 + self.bot
        smallest = min((r + 1) * self.H // self.world - r * self.H // self.world for r in range(self.world))
        if self.world > 1 and smallest < max(self.halo_top_rows, self.halo_bot_rows):
            raise ValueError(f"slabs of {smallest} rows are smaller than the halo ({self.halo_bot_rows} rows): "
                             "use fewer ranks or a smaller halo_iters")

    @property
    def rows(self) -> int:
        return self.slab1 - self.slab0

    @property
    def owned_local(self):
        """Owned rows in slab coordinates."""
        return self.top, self.top + (self.own1 - self.own0)


class HipSlabEngine:
    """The product engine: one ``DeviceTVD`` context per rank on a dedicated torch stream,
    halo rows packed into torch-owned device buffers (plain pointers into the C ABI)."""

    def __init__(self, slab_img: np.ndarray, dim: int, mu, omega, ell, boundary, device: int):
        import torch

        from .restoration.split_bregman_tvd import DeviceTVD

        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.st = DeviceTVD(slab_img.shape, slab_img.dtype, dim=dim, device=device, row_slab=boundary,
                            stream=self.stream.cuda_stream)
        self.st.set_coeffs(mu, omega, ell, slab_img)
        self.st.upload(slab_img)

    def run(self, n_iter: int, jacobi_maxiter: int = 1, isotropic: bool = False):
        with self.torch.cuda.stream(self.stream):
            # enqueue only: the stream is drained by synchronize() / download()
            return self.st.run(n_iter, jacobi_maxiter, None, eps=None, isotropic=isotropic, flags=_lib.FLAG_ASYNC)

    def halo_buffer(self, nrows: int):
        import ctypes as C

        n = C.c_size_t(0)
        _lib.check(self.st.lib.dxh_tvd_halo_bytes(self.st._h, nrows, C.byref(n)))
        return self.torch.empty(n.value, dtype=self.torch.uint8, device=self.device)

    def pack(self, row0: int, nrows: int, buf):
        _lib.check(self.st.lib.dxh_tvd_pack_rows(self.st._h, row0, nrows, buf.data_ptr()))

    def unpack(self, row0: int, nrows: int, buf):
        _lib.check(self.st.lib.dxh_tvd_unpack_rows(self.st._h, row0, nrows, buf.data_ptr()))

    def partial_norms(self, row0: int, nrows: int, want_img: bool):
        import ctypes as C

        a, b = C.c_double(0.0), C.c_double(0.0)
        _lib.check(self.st.lib.dxh_tvd_partial_norms(self.st._h, row0, nrows, C.byref(a),
                                                      C.byref(b) if want_img else None))
        return a.value, b.value

    def download(self) -> np.ndarray:
        return self.st.download()

    def synchronize(self):
        _lib.check(self.st.lib.dxh_tvd_sync(self.st._h))

    def comm_context(self):
        return self.torch.cuda.stream(self.stream)

    def close(self):
        self.st.close()


class SlabTVD:
    """Distributed split-Bregman TVD of one image cut into row slabs.

    ``slab_img`` is THIS rank's slab (rows ``plan.slab0:plan.slab1`` of the global image, halos
    included).  ``engine_factory(slab_img, dim, mu, omega, ell, boundary)`` builds the per-rank compute
    engine; the default is the HIP engine (tests inject a CPU oracle engine to exercise the exchange
    logic without a GPU)."""

    def __init__(self, plan: SlabPlan, slab_img: np.ndarray, mu=1.0, omega=1.0, ell=None, dim: int = 2,
                 halo_iters: int = 1, jacobi_maxiter: int = 1, isotropic: bool = False, engine_factory=None,
                 group=None, device: Optional[int] = None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.plan = plan
        self.k = int(halo_iters)
        self.m = int(jacobi_maxiter)
        self.isotropic = isotropic
        self.group = group
        if self.k < 1 or self.m < 1:
            raise ValueError("halo_iters and jacobi_maxiter must be >= 1")
        need_top, need_bot = self.k * self.m, self.k * self.m + 1
        if plan.world > 1 and (plan.halo_top_rows < need_top or plan.halo_bot_rows < need_bot):
            raise ValueError(f"plan halos ({plan.halo_top_rows}, {plan.halo_bot_rows}) too small for "
                             f"halo_iters={self.k}, jacobi_maxiter={self.m}: need ({need_top}, {need_bot})")
        if slab_img.shape[0] != plan.rows:
            raise ValueError(f"slab has {slab_img.shape[0]} rows, plan says {plan.rows}")
        for name, c in (("mu", mu), ("omega", omega), ("ell", ell)):
            if isinstance(c, np.ndarray) and c.ndim > 0 and c.shape[0] != plan.rows:
                raise ValueError(f"array-valued {name} must be given per slab ({plan.rows} rows)")
        boundary = (plan.rank == 0, plan.rank == plan.world - 1)
        if engine_factory is None:
            if device is None:
                device = _lib.default_device()
            engine_factory = lambda *a: HipSlabEngine(*a, device=device)  # noqa: E731
        self.engine = engine_factory(slab_img, dim, mu, omega, ell, boundary)
        self.world, self.rank = plan.world, plan.rank
        if self.world > 1:
            p = plan
            # what I send up / receive from above, what I send down / receive from below (slab rows)
            self._buf = {}
            if self.rank > 0:
                self._buf["send_up"] = self.engine.halo_buffer(p.halo_bot_rows)   # my first rows -> upper's bottom halo
                self._buf["recv_up"] = self.engine.halo_buffer(p.halo_top_rows)   # upper's last rows -> my top halo
            if self.rank < self.world - 1:
                self._buf["send_dn"] = self.engine.halo_buffer(p.halo_top_rows)   # my last rows -> lower's top halo
                self._buf["recv_dn"] = self.engine.halo_buffer(p.halo_bot_rows)   # lower's first rows -> my bottom halo
        self.iterations = 0
        self._img_norm = None

    # ---- halo exchange ---------------------------------------------------------------------------
    def exchange(self):
        if self.world == 1:
            return
        dist, p, eng = self.dist, self.plan, self.engine
        o0, o1 = p.owned_local
        backend = dist.get_backend(self.group)
        with eng.comm_context():
            if self.rank > 0:
                eng.pack(o0, p.halo_bot_rows, self._buf["send_up"])
            if self.rank < self.world - 1:
                eng.pack(o1 - p.halo_top_rows, p.halo_top_rows, self._buf["send_dn"])
            stage = backend == "gloo" and any(getattr(b, "is_cuda", False) for b in self._buf.values())
            bufs = {k: (v.cpu() if stage else v) for k, v in self._buf.items()}
            ops = []
            if self.rank > 0:
                ops.append(dist.P2POp(dist.isend, bufs["send_up"], self._peer(self.rank - 1), self.group))
                ops.append(dist.P2POp(dist.irecv, bufs["recv_up"], self._peer(self.rank - 1), self.group))
            if self.rank < self.world - 1:
                ops.append(dist.P2POp(dist.isend, bufs["send_dn"], self._peer(self.rank + 1), self.group))
                ops.append(dist.P2POp(dist.irecv, bufs["recv_dn"], self._peer(self.rank + 1), self.group))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if stage:
                for k in ("recv_up", "recv_dn"):
                    if k in bufs:
                        self._buf[k].copy_(bufs[k])
            if self.rank > 0:
                eng.unpack(0, p.halo_top_rows, self._buf["recv_up"])
            if self.rank < self.world - 1:
                eng.unpack(o1, p.halo_bot_rows, self._buf["recv_dn"])

    def _peer(self, r: int) -> int:
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    # ---- iteration -------------------------------------------------------------------------------
    def run(self, n_iter: int, eps: Optional[float] = None) -> int:
        """Run up to ``n_iter`` outer iterations; returns the number done (all ranks agree)."""
        done = 0
        step = 1 if eps is not None else self.k
        o0, o1 = self.plan.owned_local
        while done < n_iter:
            k = min(step, n_iter - done)
            self.engine.run(k, self.m, self.isotropic)
            done += k
            self.iterations += k
            if eps is not None:
                inc, img2 = self.engine.partial_norms(o0, o1 - o0, self._img_norm is None)
                if self._img_norm is None:
                    t = self.torch.tensor([inc, img2], dtype=self.torch.float64)
                else:
                    t = self.torch.tensor([inc, 0.0], dtype=self.torch.float64)
                if self.world > 1:
                    t = self._allreduce_sum(t)
                if self._img_norm is None:
                    self._img_norm = math.sqrt(float(t[1]))
                rel = math.sqrt(float(t[0])) / self._img_norm
                if rel < eps:
                    break
            if done < n_iter:
                self.exchange()
        return done

    def _allreduce_sum(self, t):
        dist = self.dist
        if dist.get_backend(self.group) == "nccl":
            t = t.to(self.engine.device)
            dist.all_reduce(t, group=self.group)
            return t.cpu()
        dist.all_reduce(t, group=self.group)
        return t

    def owned_result(self) -> np.ndarray:
        o0, o1 = self.plan.owned_local
        return self.engine.download()[o0:o1]

    def synchronize(self):
        self.engine.synchronize()

    def close(self):
        self.engine.close()


def make_plan(H: int, world: int, rank: int, halo_iters: int = 1, jacobi_maxiter: int = 1) -> SlabPlan:
    km = halo_iters * jacobi_maxiter
    return SlabPlan(H, world, rank, halo_top_rows=km, halo_bot_rows=km + 1)
