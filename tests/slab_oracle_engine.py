"""Test-only pieces for the row-slab driver: a slab engine that steps the NumPy oracle over one row slab
(halos included) as if it were a stand-alone image, and a gloo transport (torch.distributed lives in the
tests only; the product is torch-free).  They let the decomposition and halo-exchange logic run on CPU."""
import numpy as np

from oracle import tvd_numpy as O


class OracleSlabEngine:
    def __init__(self, slab_img, dim, mu, omega, ell, boundary):
        self.img = slab_img
        self.dim = dim
        self.kw = dict(mu=mu, omega=omega, ell=ell, dim=dim)
        self.u = slab_img.copy()
        self.d = [np.zeros_like(slab_img) for _ in range(dim)]
        self.b = [np.zeros_like(slab_img) for _ in range(dim)]
        self.u_prev = None

    def run(self, n_iter, jacobi_maxiter=1, isotropic=False):
        for _ in range(n_iter):
            self.u_prev = self.u
            self.u, self.d, self.b, _ = O.split_bregman_tvd(
                self.img, max_num_iter=1, x0=(self.u, self.d, self.b), isotropic=isotropic,
                jacobi_maxiter=jacobi_maxiter, promotion="strict", return_state=True, **self.kw)

    def _fields(self):
        return [self.u] + self.d + self.b

    def halo_buffer(self, nrows):
        n = len(self._fields()) * nrows * int(np.prod(self.img.shape[1:])) * self.img.dtype.itemsize
        return np.empty(n, dtype=np.uint8)

    def pack(self, row0, nrows):
        raw = b"".join(np.ascontiguousarray(f[row0:row0 + nrows]).tobytes() for f in self._fields())
        return np.frombuffer(bytearray(raw), dtype=np.uint8)

    def unpack(self, row0, nrows, buf):
        raw = np.asarray(buf).tobytes()
        per = len(raw) // len(self._fields())
        self.u = self.u.copy()
        self.d = [a.copy() for a in self.d]
        self.b = [a.copy() for a in self.b]
        for i, f in enumerate(self._fields()):
            blk = np.frombuffer(raw[i * per:(i + 1) * per], dtype=self.img.dtype).reshape((nrows,) + self.img.shape[1:])
            f[row0:row0 + nrows] = blk

    def partial_norms(self, row0, nrows, want_img):
        inc = ((self.u[row0:row0 + nrows].astype(np.float64) - self.u_prev[row0:row0 + nrows]) ** 2).sum()
        img2 = (self.img[row0:row0 + nrows].astype(np.float64) ** 2).sum() if want_img else 0.0
        return float(inc), float(img2)

    def download(self):
        return self.u

    def synchronize(self):
        pass

    def close(self):
        pass


class GlooTransport:
    """The host-transport protocol of darsia_amd.parallel over an initialised gloo process group."""

    def __init__(self):
        import torch.distributed as dist

        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def exchange(self, ops):
        import torch

        dist = self.dist
        p2p, back = [], []
        for peer, is_send, buf in ops:
            a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
            if is_send:
                t = torch.from_numpy(np.array(a, copy=True))
                p2p.append(dist.P2POp(dist.isend, t, peer))
            else:
                t = torch.empty(a.size, dtype=torch.uint8)
                p2p.append(dist.P2POp(dist.irecv, t, peer))
                back.append((a, t))
        for w in dist.batch_isend_irecv(p2p):
            w.wait()
        for a, t in back:
            a[:] = t.numpy()

    def allreduce(self, vals, op):
        import torch

        t = torch.from_numpy(np.array(vals, dtype=np.float64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == 0 else self.dist.ReduceOp.MAX)
        vals[:] = t.numpy()
