"""Test-only slab engine: the NumPy oracle stepping one row slab (halos included)
as if it were a stand-alone image.  Lets the distributed driver's decomposition
and halo-exchange logic run on CPU (gloo) without any HIP code."""
import contextlib

import numpy as np
import torch

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
        self.device = torch.device("cpu")

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
        return torch.empty(n, dtype=torch.uint8)

    def pack(self, row0, nrows, buf):
        raw = b"".join(np.ascontiguousarray(f[row0:row0 + nrows]).tobytes() for f in self._fields())
        buf.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))

    def unpack(self, row0, nrows, buf):
        raw = buf.numpy().tobytes()
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

    def comm_context(self):
        return contextlib.nullcontext()

    def close(self):
        pass
