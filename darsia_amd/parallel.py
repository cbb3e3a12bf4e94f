"""Row-slab decomposition of one TV problem over several GPUs: one process per
GPU, halo rows exchanged between slab neighbours by ``libdarsia_hip`` itself
(grouped RCCL send/recv over xGMI, started by a device-side counter and
overlapped with the interior rows -- ``dxh_tvd_run_slab`` in
``include/darsia_hip.h``).  No PyTorch anywhere:
Python only bootstraps the ranks (``RANK`` / ``WORLD_SIZE`` / ``LOCAL_RANK`` from
the launcher's environment, the 128-byte RCCL id through a file).

The reference is single-process, so nothing here mirrors
reference code; what must hold is that the N-rank result equals the 1-rank
result BIT FOR BIT (same arithmetic per pixel) -- ``tests/test_parallel_cpu.py``
(CPU, oracle engine) and ``tests/test_parallel_gpu.py`` (HIP engine) check it.

Decomposition.  The image is cut along axis 0 into contiguous slabs.  One outer
iteration with ``m`` Jacobi sweeps reads one row above and below per sweep
(utils/linear_solvers/jacobi.py:43-54) and the split variables one row above
(restoration/split_bregman_tvd.py:84-89); the fused kernel additionally needs
the new iterate one row below (:144).  A slab edge that is not the image
boundary therefore loses ``m`` valid rows per iteration at the top and ``m``
(+1 once) at the bottom.  With halos of ``k*m`` rows above and ``k*m + 1``
below, ``k`` iterations run between two exchanges; the halos are recomputed
redundantly, which costs (2km+1)/rows extra work and turns 3 small messages per
iteration into one larger one every ``k`` iterations (xGMI is point to point:
each slab talks to its two neighbours only, no ring, no collective on the data
path).

A global ``eps`` stopping test needs the increment norm every iteration: one
8-byte all-reduce per iteration is added, as in the reference's per-iteration
test (split_bregman_tvd.py:161-175).

Transports.  ``Comm.rccl`` is the product path.  ``Comm.host`` stages the halo
rows through pinned host memory and hands them to a Python transport object
(``SocketTransport`` here, a gloo one in the tests): for rehearsals and tests on
boxes with fewer GPUs than ranks, which RCCL refuses.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import socket
import struct
import tempfile
import threading
import time
from dataclasses import dataclass
from typing import Optional, Sequence

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
    periodic: bool = False   # world == 1 only: the image is periodic along axis 0, the slab is its own neighbour

    def __post_init__(self):
        if self.world < 1 or not (0 <= self.rank < self.world):
            raise ValueError("bad rank/world")
        if self.periodic and self.world != 1:
            raise ValueError("periodic slabs are a single-rank mode")
        self.own0 = self.rank * self.H // self.world
        self.own1 = (self.rank + 1) * self.H // self.world
        self.top = 0 if (self.rank == 0 and not self.periodic) else self.halo_top_rows
        self.bot = 0 if (self.rank == self.world - 1 and not self.periodic) else self.halo_bot_rows
        self.slab0 = self.own0 - self.top
        self.slab1 = self.own1 + self.bot
        smallest = min((r + 1) * self.H // self.world - r * self.H // self.world for r in range(self.world))
        if (self.world > 1 or self.periodic) and smallest < max(self.halo_top_rows, self.halo_bot_rows):
            raise ValueError(f"slabs of {smallest} rows are smaller than the halo ({self.halo_bot_rows} rows): "
                             "use fewer ranks or a smaller halo_iters")

    @property
    def rows(self) -> int:
        return self.slab1 - self.slab0

    @property
    def owned_local(self):
        """Owned rows in slab coordinates."""
        return self.top, self.top + (self.own1 - self.own0)


def make_plan(H: int, world: int, rank: int, halo_iters: int = 1, jacobi_maxiter: int = 1,
              periodic: bool = False) -> SlabPlan:
    km = halo_iters * jacobi_maxiter
    return SlabPlan(H, world, rank, halo_top_rows=km, halo_bot_rows=km + 1, periodic=periodic)


def periodic_slab(img: np.ndarray, plan: SlabPlan) -> np.ndarray:
    """The slab of a periodic single-rank plan: the image with its last rows above and its first rows below."""
    return np.ascontiguousarray(np.concatenate([img[img.shape[0] - plan.top:], img, img[:plan.bot]], axis=0))


# ---- host transports (rehearsals, tests) ---------------------------------------------------------------------------
class SocketTransport:
    """Neighbour-to-neighbour TCP on localhost: rank r listens on ``base_port + r`` and dials rank r - 1, so every
    slab interface is one bidirectional connection.  ``exchange`` moves all messages of one call together (sends from
    helper threads); ``allreduce`` folds the values up the chain of ranks and passes the result back down."""

    def __init__(self, rank: int, world: int, base_port: int, host: str = "127.0.0.1", timeout: float = 120.0):
        self.rank, self.world = rank, world
        self.conn = {}
        listener = None
        if rank < world - 1:
            listener = socket.socket()
            listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            listener.bind((host, base_port + rank))
            listener.listen(1)
        if rank > 0:
            t0 = time.time()
            while True:
                try:
                    s = socket.create_connection((host, base_port + rank - 1), timeout=timeout)
                    break
                except OSError:
                    if time.time() - t0 > timeout:
                        raise
                    time.sleep(0.05)
            s.settimeout(timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self.conn[rank - 1] = s
        if listener is not None:
            listener.settimeout(timeout)
            s, _ = listener.accept()
            s.settimeout(timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self.conn[rank + 1] = s
            listener.close()

    @staticmethod
    def _recv_into(sock, view):
        got = 0
        while got < len(view):
            n = sock.recv_into(view[got:])
            if n == 0:
                raise ConnectionError("peer closed the connection")
            got += n

    def exchange(self, ops):
        """ops: list of (peer, is_send, buffer) with writable uint8 memoryviews / arrays."""
        by_peer_send = {}
        for peer, is_send, buf in ops:
            if is_send:
                by_peer_send.setdefault(peer, []).append(buf)
        errors = []

        def send_all(peer, bufs):
            try:
                for b in bufs:
                    self.conn[peer].sendall(memoryview(b).cast("B"))
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=send_all, args=(p, b)) for p, b in by_peer_send.items()]
        for t in threads:
            t.start()
        for peer, is_send, buf in ops:
            if not is_send:
                self._recv_into(self.conn[peer], memoryview(buf).cast("B"))
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    def allreduce(self, vals: np.ndarray, op: int):
        """In place over all ranks; op 0 = sum, 1 = max.  The fold order is fixed (rank 0 upwards), so every rank gets
        the same bits."""
        n = len(vals)
        fmt = f"<{n}d"
        acc = np.array(vals, dtype=np.float64)
        if self.rank > 0:
            raw = bytearray(8 * n)
            self._recv_into(self.conn[self.rank - 1], memoryview(raw))
            below = np.array(struct.unpack(fmt, raw))
            acc = below + acc if op == 0 else np.maximum(below, acc)
        if self.rank < self.world - 1:
            self.conn[self.rank + 1].sendall(struct.pack(fmt, *acc))
            raw = bytearray(8 * n)
            self._recv_into(self.conn[self.rank + 1], memoryview(raw))
            acc = np.array(struct.unpack(fmt, raw))
        if self.rank > 0:
            self.conn[self.rank - 1].sendall(struct.pack(fmt, *acc))
        vals[:] = acc

    def close(self):
        for s in self.conn.values():
            try:
                s.close()
            except OSError:
                pass
        self.conn = {}


# ---- communicators (dxh_comm_t) ------------------------------------------------------------------------------------
_rccl_seq = 0   # communicators this process has created: every rank creates them in the same order


def _rendezvous_path() -> str:
    """One file per communicator: ranks of one launch share the launcher as parent and the rendezvous port, and the n-th
    communicator of every rank gets the suffix n -- a second ``Comm.rccl`` in the same launch can never read the id of
    the first."""
    global _rccl_seq
    seq = _rccl_seq
    _rccl_seq += 1
    explicit = os.environ.get("DXH_RENDEZVOUS_FILE")
    if explicit:
        return f"{explicit}.{seq}" if seq else explicit
    tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{seq}"
    return os.path.join(tempfile.gettempdir(), f"dxh_rccl_{tag}.id")


class Comm:
    """A ``dxh_comm_t``: RCCL (``Comm.rccl``) or host-staged through a Python transport (``Comm.host``)."""

    def __init__(self, handle, rank, world, device, keep=(), cleanup=None):
        self.lib = _lib.load()
        self._h = handle
        self.rank, self.world, self.device = rank, world, device
        self._keep = keep          # callback objects must outlive the communicator
        self._cleanup = cleanup

    @classmethod
    def rccl(cls, rank: int, world: int, device: int, timeout: float = 300.0) -> "Comm":
        """Rank 0 draws the RCCL unique id and publishes it through a file (atomic rename); the others poll for it."""
        lib = _lib.load()
        _lib.require_gpu()
        path = _rendezvous_path()
        buf = (C.c_char * _lib.COMM_ID_BYTES)()
        if rank == 0:
            _lib.check(lib.dxh_comm_unique_id(buf))
            # written to a private temporary (O_EXCL, mode 0600) and renamed into place: a leftover of a crashed run is
            # replaced atomically, never appended to or half read.  (A leftover can only be READ by a rank that gets here
            # before rank 0 publishes AND shares launcher pid, port, run id and sequence number with the crashed run --
            # i.e. with an explicit DXH_RENDEZVOUS_FILE; such a caller removes the file before launching.)
            tmp = f"{path}.{os.getpid()}.tmp"
            try:
                os.unlink(tmp)
            except OSError:
                pass
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
            with os.fdopen(fd, "wb") as f:
                f.write(bytes(buf))
            os.replace(tmp, path)
        else:
            t0 = time.time()
            while True:
                try:
                    with open(path, "rb") as f:
                        raw = f.read()
                    if len(raw) == _lib.COMM_ID_BYTES:
                        break
                except FileNotFoundError:
                    pass
                if time.time() - t0 > timeout:
                    raise TimeoutError(f"rank {rank}: no RCCL id at {path} after {timeout:.0f} s (is rank 0 running?)")
                time.sleep(0.02)
            C.memmove(buf, raw, _lib.COMM_ID_BYTES)
        h = C.c_void_p()
        _lib.check(lib.dxh_comm_create_rccl(C.byref(h), device, buf, rank, world))

        comm = cls(h, rank, world, device)
        comm.barrier()   # every rank has read the id (ncclCommInitRank has returned everywhere)
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
        return comm

    @classmethod
    def host(cls, transport, device: int) -> "Comm":
        """Host-staged communicator over ``transport`` (``rank``, ``world``, ``exchange(ops)``, ``allreduce(vals, op)``)."""
        lib = _lib.load()
        _lib.require_gpu()
        failure = []

        def xfn(user, n, peers, is_send, bufs, nbytes):
            try:
                ops = []
                for i in range(n):
                    arr = (C.c_ubyte * nbytes[i]).from_address(bufs[i])
                    ops.append((peers[i], bool(is_send[i]), arr))
                transport.exchange(ops)
                return 0
            except Exception as e:  # noqa: BLE001 -- must not propagate through the C frame
                failure.append(e)
                return 1

        def afn(user, vals, n, op):
            try:
                a = np.ctypeslib.as_array(vals, shape=(n,))
                transport.allreduce(a, op)
                return 0
            except Exception as e:  # noqa: BLE001
                failure.append(e)
                return 1

        cx, ca = _lib.COMM_EXCHANGE_FN(xfn), _lib.COMM_ALLREDUCE_FN(afn)
        h = C.c_void_p()
        _lib.check(lib.dxh_comm_create_host(C.byref(h), device, transport.rank, transport.world, cx, ca, None))
        comm = cls(h, transport.rank, transport.world, device, keep=(cx, ca, transport))
        comm.failure = failure
        return comm

    def allreduce(self, vals: Sequence[float], op: str = "sum"):
        a = (C.c_double * len(vals))(*vals)
        _lib.check(self.lib.dxh_comm_allreduce(self._h, a, len(vals), 0 if op == "sum" else 1))
        return list(a)

    def barrier(self):
        _lib.check(self.lib.dxh_comm_barrier(self._h))

    def info(self) -> dict:
        r, w, k = C.c_int(), C.c_int(), C.c_int()
        x, b = C.c_int64(), C.c_int64()
        _lib.check(self.lib.dxh_comm_info(self._h, C.byref(r), C.byref(w), C.byref(k), C.byref(x), C.byref(b)))
        return dict(rank=r.value, world=w.value, kind="rccl" if k.value == 0 else "host", exchanges=x.value,
                    bytes_sent=b.value)

    def close(self):
        if self._h:
            self.lib.dxh_comm_destroy(self._h)
            self._h = None
            if self._cleanup:
                self._cleanup()

    __del__ = close


def init_from_env(device: Optional[int] = None, rehearsal_base_port: Optional[int] = None):
    """(comm, rank, world, local_rank) from the launcher's environment (``torchrun`` / ``torch.distributed.run`` set
    RANK, WORLD_SIZE, LOCAL_RANK, MASTER_PORT; the launcher is only a process spawner here).  ``rehearsal_base_port``:
    host-staged ``SocketTransport`` instead of RCCL, ranks round-robin over the visible GPUs."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if rehearsal_base_port is not None:
        dev = local_rank % max(1, _lib.device_count()) if device is None else device
        return Comm.host(SocketTransport(rank, world, rehearsal_base_port), dev), rank, world, dev
    if device is None:
        ndev = _lib.device_count()
        masked = any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
        if local_rank < ndev:
            dev = local_rank
        elif ndev == 1 and masked:
            dev = 0          # the launcher handed every rank its own GPU through a visibility mask
        else:
            raise _lib.HipError(f"rank {rank}: LOCAL_RANK {local_rank} but {ndev} visible GPU(s) -- one process per GPU is "
                                "needed for RCCL (rehearse on fewer GPUs with the host-staged transport: "
                                "DXH_BENCH_REHEARSAL=1 / init_from_env(rehearsal_base_port=...))")
    else:
        dev = device
    return Comm.rccl(rank, world, dev), rank, world, dev


def slab_promotion(promotion: Optional[str], dtype, mu, omega, ell, dim: int, jacobi_maxiter: int, isotropic: bool) -> str:
    """The dtype flow a slab run takes -- ``split_bregman_tvd``'s policy (``choose_promotion``) with what the slab path of
    the numpy2 flow covers (``dxh_tvd_run_slab``: dim = 2, one Jacobi sweep, anisotropic, scalar omega and ell; mu a
    scalar or a float64 array)."""
    from .restoration.split_bregman_tvd import choose_promotion

    why_not = None
    if np.dtype(dtype) != np.float32:
        why_not = "the image is not float32"
    elif dim != 2 or jacobi_maxiter != 1 or isotropic:
        why_not = "row slabs: dim = 2, one Jacobi sweep, anisotropic only"
    elif np.ndim(omega) != 0 or (ell is not None and np.ndim(ell) != 0) or (ell is None and np.ndim(mu) != 0):
        why_not = "row slabs: scalar omega and ell only"
    elif np.ndim(mu) != 0 and np.asarray(mu).dtype != np.float64:
        why_not = "row slabs: mu a scalar or a float64 array"
    chosen = choose_promotion(promotion, dtype, (mu, omega, ell), why_not)
    if chosen == "numpy2" and why_not is not None:
        raise NotImplementedError(f"promotion='numpy2' on row slabs: {why_not}")
    return chosen


# ---- engines ---------------------------------------------------------------------------------------------------------
class HipSlabEngine:
    """The product engine: one ``DeviceTVD`` context per rank with a communicator attached; the whole round loop
    (launches, exchanges, overlap) runs inside ``dxh_tvd_run_slab``."""

    def __init__(self, slab_img: np.ndarray, dim: int, mu, omega, ell, boundary, device: int, comm: Comm,
                 plan: SlabPlan, promotion: str = "native"):
        from .restoration.split_bregman_tvd import DeviceTVD

        self.comm = comm
        self.st = DeviceTVD(slab_img.shape, slab_img.dtype, dim=dim, device=device, row_slab=boundary, promotion=promotion)
        self.st.set_coeffs(mu, omega, ell, slab_img)
        self.st.upload(slab_img)
        _lib.check(self.st.lib.dxh_tvd_attach_comm(self.st._h, comm._h, plan.halo_top_rows, plan.halo_bot_rows))

    def run_slab(self, n_iter: int, jacobi_maxiter: int, isotropic: bool, eps: Optional[float], flags: int = 0):
        res = _lib.TvdResult()
        if isotropic:
            flags |= _lib.FLAG_ISOTROPIC
        _lib.check(self.st.lib.dxh_tvd_run_slab(self.st._h, int(n_iter), int(jacobi_maxiter),
                                                 -1.0 if eps is None else float(eps), flags, C.byref(res)))
        failure = getattr(self.comm, "failure", None)
        if failure:
            raise failure[0]
        return res

    def exchange(self):
        _lib.check(self.st.lib.dxh_tvd_exchange(self.st._h))

    def download(self) -> np.ndarray:
        return self.st.download()

    def synchronize(self):
        _lib.check(self.st.lib.dxh_tvd_sync(self.st._h))

    def stats(self) -> dict:
        st = _lib.SlabStats()
        st.struct_size = C.sizeof(_lib.SlabStats)
        _lib.check(self.st.lib.dxh_tvd_slab_stats(self.st._h, C.byref(st)))
        return {name: getattr(st, name) for name, _ in _lib.SlabStats._fields_ if name != "struct_size"}

    def close(self):
        self.st.close()


class SlabTVD:
    """Distributed split-Bregman TVD of one image cut into row slabs.

    ``slab_img`` is THIS rank's slab (rows ``plan.slab0:plan.slab1`` of the global image, halos
    included).  Product use: pass ``comm`` (``Comm.rccl`` / ``init_from_env``), the HIP engine does
    everything inside ``libdarsia_hip``.  Tests inject ``engine_factory(slab_img, dim, mu, omega, ell,
    boundary)`` (a CPU oracle engine) together with a host ``transport`` to exercise the decomposition and
    exchange logic without a GPU: that path runs the round loop below in Python."""

    def __init__(self, plan: SlabPlan, slab_img: np.ndarray, mu=1.0, omega=1.0, ell=None, dim: int = 2,
                 halo_iters: int = 1, jacobi_maxiter: int = 1, isotropic: bool = False, engine_factory=None,
                 comm: Optional[Comm] = None, transport=None, device: Optional[int] = None, flags: int = 0,
                 promotion: Optional[str] = None):
        self.plan = plan
        self.k = int(halo_iters)
        self.m = int(jacobi_maxiter)
        self.isotropic = isotropic
        self.flags = flags
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
        boundary = (plan.top == 0, plan.bot == 0)
        self.world, self.rank = plan.world, plan.rank
        self.transport = transport
        if engine_factory is None:
            if comm is None:
                raise ValueError("SlabTVD needs a communicator (Comm.rccl / Comm.host / init_from_env)")
            if (comm.rank, comm.world) != (plan.rank, plan.world):
                raise ValueError("communicator and plan disagree on rank / world")
            if device is None:
                device = comm.device
            # the dtype flow of a float32 image follows split_bregman_tvd's policy: an explicit ``promotion`` wins, then
            # DXH_PROMOTION, then the reference's own flow ("numpy2", restoration/split_bregman_tvd.py under NumPy >= 2)
            # wherever the slab path has it, the native float32 flow otherwise (the log says why)
            self.promotion = slab_promotion(promotion, slab_img.dtype, mu, omega, ell, dim, self.m, isotropic)
            self.engine = HipSlabEngine(slab_img, dim, mu, omega, ell, boundary, device, comm, plan, self.promotion)
        else:
            if plan.world > 1 and transport is None:
                raise ValueError("an injected engine needs a host transport")
            self.engine = engine_factory(slab_img, dim, mu, omega, ell, boundary)
            self._valid = plan.halo_top_rows   # halo rows still valid (uploaded slabs carry valid halos)
        self.iterations = 0
        self._img_norm = None

    # ---- host-side round loop (injected engines only) -----------------------------------------------------------
    def _exchange_host(self):
        p, eng, tr = self.plan, self.engine, self.transport
        o0, o1 = p.owned_local
        ops, recvs = [], []
        if self.rank > 0:
            ops.append((self.rank - 1, True, eng.pack(o0, p.halo_bot_rows)))          # my first rows -> upper's bottom halo
            buf = eng.halo_buffer(p.halo_top_rows)
            ops.append((self.rank - 1, False, buf))                                     # upper's last rows -> my top halo
            recvs.append((0, p.halo_top_rows, buf))
        if self.rank < self.world - 1:
            ops.append((self.rank + 1, True, eng.pack(o1 - p.halo_top_rows, p.halo_top_rows)))  # -> lower's top halo
            buf = eng.halo_buffer(p.halo_bot_rows)
            ops.append((self.rank + 1, False, buf))                                     # lower's first rows -> my bottom halo
            recvs.append((o1, p.halo_bot_rows, buf))
        tr.exchange(ops)
        for row0, nrows, buf in recvs:
            eng.unpack(row0, nrows, buf)
        self._valid = p.halo_top_rows

    def _run_host(self, n_iter: int, eps: Optional[float]) -> int:
        done = 0
        o0, o1 = self.plan.owned_local
        while done < n_iter:
            if self.world > 1 and self._valid < self.m:
                self._exchange_host()
            left = n_iter - done
            k = 1 if eps is not None else (min(left, self._valid // self.m) if self.world > 1 else left)
            self.engine.run(k, self.m, self.isotropic)
            done += k
            self._valid -= k * self.m
            if eps is not None:
                inc, img2 = self.engine.partial_norms(o0, o1 - o0, self._img_norm is None)
                vals = np.array([inc, img2 if self._img_norm is None else 0.0])
                if self.world > 1:
                    self.transport.allreduce(vals, 0)
                if self._img_norm is None:
                    self._img_norm = math.sqrt(float(vals[1]))
                if math.sqrt(float(vals[0])) / self._img_norm < eps:
                    break
        return done

    # ---- iteration -------------------------------------------------------------------------------------------------
    def run(self, n_iter: int, eps: Optional[float] = None, more_to_come: bool = False, asynchronous: bool = False):
        """Run up to ``n_iter`` outer iterations; returns the number done (all ranks agree).  ``more_to_come``: the
        halos are refreshed at the end (overlapped with the last launch) so that a following ``run`` starts at once."""
        if hasattr(self.engine, "run_slab"):
            flags = self.flags | (_lib.FLAG_CONTINUE if more_to_come else 0) | (_lib.FLAG_ASYNC if asynchronous else 0)
            res = self.engine.run_slab(n_iter, self.m, self.isotropic, eps, flags)
            self.last_result = res
            done = res.iters_done
        else:
            done = self._run_host(n_iter, eps)
        self.iterations += done
        return done

    def exchange(self):
        """Refresh the halos now (normally ``run`` decides when)."""
        if self.world == 1:
            return
        if hasattr(self.engine, "run_slab"):
            self.engine.exchange()
        else:
            self._exchange_host()

    def owned_result(self) -> np.ndarray:
        o0, o1 = self.plan.owned_local
        return self.engine.download()[o0:o1]

    def synchronize(self):
        self.engine.synchronize()

    def slab_stats(self) -> dict:
        """Launch / exchange counts and where the exchange stream's time went (``dxh_tvd_slab_stats``); HIP engine only."""
        return self.engine.stats()

    def close(self):
        self.engine.close()
