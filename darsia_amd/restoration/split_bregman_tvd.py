"""Split-Bregman TV denoising on MI355X (reference: restoration/split_bregman_tvd.py:14-177).

Same signature and semantics as ``darsia.split_bregman_tvd``; the iteration
loop runs in ``libdarsia_hip`` (fused rhs + Jacobi + shrink sweep, see
``darsia_amd/csrc/dxh_kernels.h``).  The image stays on the device between
iterations; only the result (and, for a warm start, the split variables) comes
back.

Divergences from the reference, all deliberate:

* ``solver=None`` creates a fresh ``Jacobi()`` per call instead of sharing one
  def-time instance whose cached diagonal never updates (quirks Q1, Q3);
* two dtype flows for float32 images (``promotion``, an extra keyword; ``DXH_PROMOTION`` sets the
  default), each the reference's own arithmetic under one NumPy generation, each reproduced BIT FOR BIT.
  ``"numpy2"`` -- THE DEFAULT wherever it exists (the ``Jacobi`` solver, anisotropic; DESIGN.md
  section 3 lists the rest) -- is the reference under NumPy >= 2: the Jacobi iterate is float64
  (jacobi.py:101-114), ``d``, ``b`` and the right-hand side float32, a float64 weight map the
  threshold (``PipeNp2`` in dxh_np2_pipe.h, ``k_fused_np2``; 0.285 ms per iteration at 4000 x 8000 x 3).
  ``"native"`` keeps every quantity float32 (0.15 ms): the reference under NumPy < 2 (value-based
  casting), pinned on vectors made under NumPy 1.26 (tests/test_np1_golden_gpu.py).  The two are 2e-7
  apart while the shrinkage is inactive (``mu / ell`` above the image's gradients: ``mu = 0.1``,
  ``ell = 2 mu`` on a [0, 1] image, the benchmark's setting) and 1e-4 absolute = 10 - 80 x rtol 1e-4 once
  it is active (``mu = 1``, or ``ell >= 0.3`` at ``mu = 0.1``: a near-threshold element that shrinks
  in one flow and not in the other).  float64 images have one flow, which is the reference's, bit for bit;
* the isotropic variant also accepts float32 and colour images (the reference's
  numba kernel is float64 / 2-D only);
* float images only (integer images silently break the reference, quirk Q5).
"""
from __future__ import annotations

import ctypes as C
import logging
import os
import threading
from typing import Optional, Union

import numpy as np

from .. import _lib
from ..device import DeviceArray, host_empty
from ..utils.linear_solvers.cg import CG
from ..utils.linear_solvers.jacobi import Jacobi
from ..utils.linear_solvers.mg import MG
from ..utils.linear_solvers.solver import Solver

logger = logging.getLogger(__name__)


def _coef(value, img, name: str, keep_float_type: bool = False):
    """(scalar, array or None).  ``keep_float_type``: a float32 / float64 array stays in its own type -- the library
    converts it on the device (``dxh_tvd_set_coeffs_typed``) instead of a host pass over it."""
    if isinstance(value, DeviceArray):
        ok_type = value.dtype in (np.float32, np.float64) if keep_float_type else value.dtype == img.dtype
        if value.shape != tuple(img.shape) or not ok_type:
            raise ValueError(f"a device-resident {name} must have the image's shape and " +
                             ("a float dtype" if keep_float_type else "dtype"))
        return 0.0, value
    if isinstance(value, np.ndarray) and value.ndim > 0:
        try:
            arr = np.broadcast_to(value, img.shape)
        except ValueError as e:
            raise ValueError(f"{name} is not broadcastable to the image shape {img.shape}") from e
        if keep_float_type and arr.dtype in (np.float32, np.float64):
            return 0.0, np.ascontiguousarray(arr)
        return 0.0, np.ascontiguousarray(arr, dtype=img.dtype)
    return float(value), None


class DeviceTVD:
    """Device-resident split-Bregman state of one image (thin wrapper of ``dxh_tvd_*``).

    ``row_slab=(top_is_boundary, bottom_is_boundary)`` describes a slab of a larger
    image (multi-GPU decomposition, ``darsia_amd.parallel``)."""

    def __init__(self, shape, dtype, dim: int = 2, device: Optional[int] = None, nsum: int = _lib.NSUM_ALL_AXES,
                 row_slab=(True, True), stream: Optional[int] = None, promotion: str = "native"):
        self.lib = _lib.load()
        _lib.require_gpu()
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)          # the image's dtype: what upload takes and download returns
        if promotion not in ("native", "numpy2"):
            raise ValueError(f"promotion must be 'native' or 'numpy2', not {promotion!r}")
        self.numpy2 = promotion == "numpy2"
        if self.numpy2 and self.dtype != np.float32:
            raise ValueError("promotion='numpy2' is the reference's flow for float32 images (float64 images run natively "
                             "and match the reference bit for bit as they are)")
        self.ctx_dtype = np.dtype(np.float64) if self.numpy2 else self.dtype
        self.dim = dim
        ndim, n0, n1, n2 = _lib.shape3(self.shape, dim)
        d = _lib.TvdDesc()
        d.struct_size = C.sizeof(_lib.TvdDesc)
        d.device = _lib.default_device() if device is None else device
        self.device = int(d.device)
        d.dtype = _lib.dtype_code(self.ctx_dtype)
        d.ndim, d.dim, d.nsum = ndim, dim, nsum
        d.n0, d.n1, d.n2 = n0, n1, n2
        d.top_is_boundary, d.bottom_is_boundary = int(row_slab[0]), int(row_slab[1])
        d.stream = stream
        self._h = C.c_void_p()
        _lib.check(self.lib.dxh_tvd_create(C.byref(self._h), C.byref(d)))
        self._promotion_mode = _lib.PROMOTION_NUMPY2 if self.numpy2 else 0
        if self.numpy2:
            _lib.check(self.lib.dxh_tvd_set_promotion(self._h, self._promotion_mode))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.dxh_tvd_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_coeffs(self, mu, omega, ell, like: np.ndarray):
        if self.numpy2:
            return self._set_coeffs_numpy2(mu, omega, ell, like)
        mu_s, mu_a = _coef(mu, like, "mu", True)
        om_s, om_a = _coef(omega, like, "omega", True)
        if ell is None:
            has_ell, ell_s, ell_a = 0, 0.0, None
        else:
            ell_s, ell_a = _coef(ell, like, "ell", True)
            has_ell = 1
        code = lambda a: _lib.dtype_code(self.dtype if a is None else a.dtype)
        _lib.check(self.lib.dxh_tvd_set_coeffs_typed(self._h, mu_s, _lib.ptr(mu_a), code(mu_a), om_s, _lib.ptr(om_a),
                                                      code(om_a), has_ell, ell_s, _lib.ptr(ell_a), code(ell_a)))

    def _set_coeffs_numpy2(self, mu, omega, ell, like):
        """Scalar omega and ell with mu a scalar or an array, or float64 arrays for omega / ell; arrays on the host or on the
        device (``DeviceArray``).  A float64 mu array is divided by ell on the device in float64 (``mu / ell``,
        split_bregman_tvd.py:144: float64 array / Python float); a float32 one the way NumPy does it -- the quotient in
        float32 (``dxh_tvd_set_coeffs_typed`` forms it on the device and hands it on as the threshold itself)."""
        is_arr = lambda c: c is not None and (isinstance(c, DeviceArray) or np.ndim(c) != 0)   # noqa: E731
        dt_of = lambda c: c.dtype if isinstance(c, DeviceArray) else np.asarray(c).dtype        # noqa: E731
        f64 = _lib.dtype_code(np.dtype(np.float64))
        if self._promotion_mode != _lib.PROMOTION_NUMPY2:
            _lib.check(self.lib.dxh_tvd_set_promotion(self._h, _lib.PROMOTION_NUMPY2))
            self._promotion_mode = _lib.PROMOTION_NUMPY2
        if is_arr(omega) or is_arr(ell) or (ell is None and is_arr(mu)):
            # array-valued omega / ell (or ell = 2 mu): float64 arrays, the coefficient planes of the double context
            # (split_bregman_tvd.py:62-68, jacobi.py:57-72,101-106 in float64, as for a float64 image)
            if any(dt_of(a) != np.float64 for a in (mu, omega, ell) if is_arr(a)):
                raise NotImplementedError("promotion='numpy2': array-valued omega / ell need float64 arrays")

            def shaped(c):
                if not is_arr(c):
                    return None
                if isinstance(c, DeviceArray):
                    if c.shape != tuple(like.shape):
                        raise ValueError("a device-resident coefficient array must have the image's shape")
                    return c
                return np.ascontiguousarray(np.broadcast_to(c, like.shape), dtype=np.float64)

            sc = lambda c: 0.0 if (c is None or is_arr(c)) else float(c)   # noqa: E731
            mu_a, om_a, ell_a = shaped(mu), shaped(omega), shaped(ell)
            _lib.check(self.lib.dxh_tvd_set_coeffs_typed(self._h, sc(mu), _lib.ptr(mu_a), f64, sc(omega) if om_a is None else float("nan"),
                                                          _lib.ptr(om_a), f64, 0 if ell is None else 1, sc(ell), _lib.ptr(ell_a), f64))
            return
        mu_s, mu_a = _coef(mu, like, "mu", True)
        has_ell, ell_s = (0, 0.0) if ell is None else (1, float(ell))
        _lib.check(self.lib.dxh_tvd_set_coeffs_typed(self._h, mu_s, _lib.ptr(mu_a), _lib.dtype_code(np.dtype(np.float64) if mu_a is None else mu_a.dtype),
                                                      float(omega), None, f64, has_ell, ell_s, None, f64))

    def upload(self, img, u0=None, d0=None, b0=None):
        if self.numpy2:
            if not isinstance(img, DeviceArray):
                img = np.ascontiguousarray(img, dtype=self.dtype)
            assert tuple(img.shape) == self.shape and img.dtype == self.dtype
            if u0 is not None or d0 is not None or b0 is not None:
                # warm start (split_bregman_tvd.py:92-96): float32 d0 / b0 as the function leaves them, u0 float32 / float64;
                # host arrays or DeviceArrays (a chain of calls whose state stays in HBM)
                if d0 is None or b0 is None:
                    raise NotImplementedError("promotion='numpy2': a warm start needs d0 and b0")
                dense = lambda a: a if isinstance(a, DeviceArray) else np.ascontiguousarray(a)   # noqa: E731
                d0c, b0c = dense(d0), dense(b0)
                if d0c.dtype != np.float32 or b0c.dtype != np.float32:
                    raise NotImplementedError("promotion='numpy2': a warm start needs float32 d0 / b0")
                assert tuple(d0c.shape) == tuple(b0c.shape) == self.shape + (self.dim,)
                u0c = None
                if u0 is not None:
                    u0c = dense(u0)
                    if u0c.dtype not in (np.float32, np.float64) or tuple(u0c.shape) != self.shape:
                        raise NotImplementedError("promotion='numpy2': u0 must be a float32 or float64 array of the image's shape")
                _lib.check(self.lib.dxh_tvd_upload_warm_np2(self._h, _lib.ptr(img), _lib.ptr(u0c),
                                                             _lib.dtype_code(u0c.dtype) if u0c is not None else 0,
                                                             _lib.ptr(d0c), _lib.ptr(b0c)))
                return
            _lib.check(self.lib.dxh_tvd_upload_typed(self._h, _lib.ptr(img), _lib.dtype_code(self.dtype)))
            return

        def c(a):
            if a is None or isinstance(a, DeviceArray):
                if a is not None and a.dtype != self.dtype:
                    raise TypeError(f"device array of dtype {a.dtype} given to a {self.dtype} context")
                return a
            return np.ascontiguousarray(a, dtype=self.dtype)

        img, u0, d0, b0 = c(img), c(u0), c(d0), c(b0)
        assert img.shape == self.shape
        _lib.check(self.lib.dxh_tvd_upload(self._h, _lib.ptr(img), _lib.ptr(u0), _lib.ptr(d0), _lib.ptr(b0)))

    def run(self, n_iter, jacobi_maxiter=1, jacobi_tol=None, eps=None, isotropic=False, norm_scale=0.0, flags=0):
        res = _lib.TvdResult()
        if isotropic:
            flags |= _lib.FLAG_ISOTROPIC
        _lib.check(self.lib.dxh_tvd_run(self._h, int(n_iter), int(jacobi_maxiter),
                                        -1.0 if jacobi_tol is None else float(jacobi_tol),
                                        -1.0 if eps is None else float(eps), float(norm_scale), flags, C.byref(res)))
        return res

    def set_solver(self, solver) -> None:
        """Use ``solver`` (Jacobi / MG / CG object) for the u-update (split_bregman_tvd.py:25,133)."""
        if isinstance(solver, MG):
            args = (_lib.SOLVER_MG, solver.depth, solver.smoother_iterations, solver.maxiter,
                    -1.0 if solver.tol is None else float(solver.tol))
        elif isinstance(solver, CG):
            if solver.tol is None:
                raise TypeError("CG needs a tolerance (scipy.sparse.linalg.cg(tol=None) fails in the reference as well)")
            args = (_lib.SOLVER_CG, 0, 0, solver.maxiter, float(solver.tol))
        else:
            args = (_lib.SOLVER_JACOBI, 0, 0, 0, -1.0)
        _lib.check(self.lib.dxh_tvd_set_solver(self._h, *args))

    def adapt_ell(self) -> None:
        """ell <- 1 / max(|grad u|_1, 1e-12) per element (split_bregman_tvd.py:149-158); explicit mode only."""
        _lib.check(self.lib.dxh_tvd_adapt_ell(self._h))

    def energy(self, previous: bool = False) -> float:
        """Energy functional of split_bregman_tvd.py:71-77 of the current (or previous) iterate."""
        v = C.c_double(0.0)
        _lib.check(self.lib.dxh_tvd_energy(self._h, 1 if previous else 0, C.byref(v)))
        return v.value

    def download(self, want_split=False, out=None, split_out=None):
        """The iterate (and the split variables) as host arrays, or into ``out`` (ndarray or DeviceArray); numpy2 contexts
        write the split variables into ``split_out = (d, b)`` when given (float32 ``(..., dim)`` DeviceArrays or contiguous
        ndarrays)."""
        u = host_empty(self.shape, self.dtype) if out is None else out
        assert tuple(u.shape) == self.shape and u.dtype == self.dtype
        assert isinstance(u, DeviceArray) or u.flags.c_contiguous
        if self.numpy2:
            # convert_dtype(img_iter, float32) (split_bregman_tvd.py:177): the float64 iterate rounded to float32
            _lib.check(self.lib.dxh_tvd_download_typed(self._h, _lib.ptr(u), _lib.dtype_code(self.dtype)))
            if not want_split:
                return u
            if split_out is not None:
                d, b = split_out
                for a in (d, b):
                    assert tuple(a.shape) == self.shape + (self.dim,) and a.dtype == np.float32
                    assert isinstance(a, DeviceArray) or a.flags.c_contiguous
            else:
                d = host_empty(self.shape + (self.dim,), np.float32)   # the reference's float32 d, b
                b = host_empty(self.shape + (self.dim,), np.float32)
            _lib.check(self.lib.dxh_tvd_download_split_np2(self._h, _lib.ptr(d), _lib.ptr(b)))
            return u, d, b
        if not want_split:
            _lib.check(self.lib.dxh_tvd_download(self._h, _lib.ptr(u), None, None))
            return u
        d = host_empty(self.shape + (self.dim,), self.dtype)
        b = host_empty(self.shape + (self.dim,), self.dtype)
        _lib.check(self.lib.dxh_tvd_download(self._h, _lib.ptr(u), _lib.ptr(d), _lib.ptr(b)))
        return u, d, b


def solve_banded(img: np.ndarray, mu, omega, ell, n_iter: int, jacobi_maxiter: int = 1, isotropic: bool = False,
                 nbands: int = 0, device: Optional[int] = None, return_bands: bool = False):
    """``n_iter`` split-Bregman iterations on a host image in row bands (``dxh_tvd_solve_banded``): band b + 1 is
    uploaded and band b - 1 downloaded while band b iterates.  ``nbands = 0`` lets the library choose."""
    lib = _lib.load()
    _lib.require_gpu()
    src = np.ascontiguousarray(img)
    mu_s, mu_a = _coef(mu, src, "mu")
    om_s, om_a = _coef(omega, src, "omega")
    has_ell, ell_s, ell_a = (0, 0.0, None) if ell is None else (1, *_coef(ell, src, "ell"))
    ndim, n0, n1, n2 = _lib.shape3(src.shape, 2)
    d = _lib.TvdDesc()
    d.struct_size = C.sizeof(_lib.TvdDesc)
    d.device = _lib.default_device() if device is None else device
    d.dtype = _lib.dtype_code(src.dtype)
    d.ndim, d.dim, d.nsum = ndim, 2, _lib.NSUM_ALL_AXES
    d.n0, d.n1, d.n2 = n0, n1, n2
    d.top_is_boundary = d.bottom_is_boundary = 1
    out = host_empty(src.shape, src.dtype)
    used = C.c_int(0)
    _lib.check(lib.dxh_tvd_solve_banded(C.byref(d), _lib.ptr(src), _lib.ptr(out), mu_s, _lib.ptr(mu_a), om_s, _lib.ptr(om_a),
                                        has_ell, ell_s, _lib.ptr(ell_a), int(n_iter), int(jacobi_maxiter),
                                        _lib.FLAG_ISOTROPIC if isotropic else 0, int(nbands), C.byref(used)))
    return (out, used.value) if return_bands else out


def solve_banded_numpy2(img: np.ndarray, mu: float, omega: float, ell: Optional[float], n_iter: int, nbands: int = 0,
                        device: Optional[int] = None, return_bands: bool = False):
    """The row-band whole call for the reference's float32 dtype flow (``dxh_tvd_solve_banded_typed``): a float32 host
    image through float64 band contexts with promotion "numpy2", scalar coefficients, Jacobi(maxiter=1)."""
    lib = _lib.load()
    _lib.require_gpu()
    src = np.ascontiguousarray(img, dtype=np.float32)
    ndim, n0, n1, n2 = _lib.shape3(src.shape, 2)
    d = _lib.TvdDesc()
    d.struct_size = C.sizeof(_lib.TvdDesc)
    d.device = _lib.default_device() if device is None else device
    d.dtype = _lib.dtype_code(np.dtype(np.float64))
    d.ndim, d.dim, d.nsum = ndim, 2, _lib.NSUM_ALL_AXES
    d.n0, d.n1, d.n2 = n0, n1, n2
    d.top_is_boundary = d.bottom_is_boundary = 1
    out = host_empty(src.shape, np.float32)
    used = C.c_int(0)
    has_ell, ell_s = (0, 0.0) if ell is None else (1, float(ell))
    _lib.check(lib.dxh_tvd_solve_banded_typed(C.byref(d), _lib.ptr(src), _lib.ptr(out), _lib.dtype_code(np.dtype(np.float32)),
                                              _lib.PROMOTION_NUMPY2, float(mu), float(omega), has_ell, ell_s, int(n_iter),
                                              0, int(nbands), C.byref(used)))
    return (out, used.value) if return_bands else out


def numpy2_unsupported(img, mu, omega, ell, dim, x0, isotropic, solver, adaptive, quiet) -> Optional[str]:
    """None when the reference's NumPy >= 2 flow for a float32 image exists on the device for this call, else why not."""
    if img.dtype != np.float32:
        return "the image is not float32"
    if dim not in (2, 3) or img.ndim not in (2, 3) or dim > img.ndim:
        return "dim = 2 or 3 on an array of 2 or 3 axes only"
    is_arr = lambda c: c is not None and (isinstance(c, DeviceArray) or np.ndim(c) != 0)   # noqa: E731
    dt_of = lambda c: c.dtype if isinstance(c, DeviceArray) else np.asarray(c).dtype        # noqa: E731
    for name, c in (("mu", mu), ("omega", omega), ("ell", ell)):
        if isinstance(c, DeviceArray) and (c.shape != tuple(img.shape) or c.dtype not in (np.float32, np.float64)):
            return f"a device-resident {name} must be a float32 / float64 array of the image's shape"
    arr_ell = is_arr(omega) or is_arr(ell) or (ell is None and is_arr(mu))
    if arr_ell:
        # omega and / or ell (given, or 2 mu) an array: float64 arrays pull parts of the right-hand side into float64
        # (k_rhs_np2_het); with float32 arrays only, every quantity of the reference stays float32 -- the native flow IS
        # the reference's (native_is_the_reference_flow); a mixture of the two is not reproduced
        kinds = {np.dtype(dt_of(c)) for c in (mu, omega, ell) if is_arr(c)}
        if kinds != {np.dtype(np.float64)}:
            return "array-valued omega / ell: float64 arrays only"
    if adaptive is not None:
        # ell becomes a float32 array (split_bregman_tvd.py:149-158 on a float32 image): explicit float32 d, b on the device
        if arr_ell:
            return "the adaptive schedule needs scalar omega and ell"
        if is_arr(mu) and dt_of(mu) != np.float64:
            return "the adaptive schedule needs mu as a scalar or a float64 array"
    if x0 is not None:
        # a warm start in the reference's flow: float32 d0 / b0 (what the function itself leaves in them for a float32
        # image), u0 float32 or float64, on the host or on the device
        u0, d0, b0 = x0
        if any(not isinstance(a, (np.ndarray, DeviceArray)) for a in (u0, d0, b0)):
            return "a warm start needs arrays (ndarray or DeviceArray)"
        if d0.dtype != np.float32 or b0.dtype != np.float32 or u0.dtype not in (np.float32, np.float64):
            return "a warm start needs float32 d0 / b0 and a float32 or float64 u0"
    if isotropic:
        return "anisotropic only (the reference's isotropic branch is float64-only)"
    if not isinstance(solver, Jacobi) or solver.maxiter < 1:
        return "the Jacobi solver only"
    return None


def native_is_the_reference_flow(img, mu, omega, ell) -> bool:
    """float32 image with an array-valued omega or ell (given, or 2 mu) and float32 arrays only: the Jacobi diagonals are
    float32 arrays then (jacobi.py:57-72,101-106), no np.float64 scalar meets the iterate, and every quantity of the
    reference stays float32 -- its flow is the native one, bit for bit."""
    if img.dtype != np.float32 or any(isinstance(c, DeviceArray) for c in (mu, omega, ell)):
        return False
    if not (np.ndim(omega) != 0 or (ell is not None and np.ndim(ell) != 0) or (ell is None and np.ndim(mu) != 0)):
        return False
    return {np.asarray(c).dtype for c in (mu, omega, ell) if c is not None and np.ndim(c) != 0} == {np.dtype(np.float32)}


def choose_promotion(promotion: Optional[str], img_dtype, coefs, why_not: Optional[str]) -> str:
    """The dtype flow a call runs in: an explicit ``promotion`` wins; else ``DXH_PROMOTION``; else -- the default -- the
    reference's own flow (``"numpy2"``) for every float32 image whose configuration has it (``why_not`` is None), the
    native flow otherwise (float64 images: native IS the reference's flow).  ``coefs`` is kept for the log message."""
    if promotion is None:
        promotion = os.environ.get("DXH_PROMOTION") or None
        if promotion is None or promotion == "numpy2":
            f32 = np.dtype(img_dtype) == np.float32
            if f32 and why_not is not None:
                logger.info("float32 image: native float32 flow, not the reference's mixed float32 / float64 one (%s)", why_not)
            promotion = "numpy2" if (f32 and why_not is None) else "native"
    if promotion not in ("native", "numpy2"):
        raise ValueError(f"promotion must be None, 'native' or 'numpy2', not {promotion!r}")
    return promotion


def split_bregman_tvd(
    img: np.ndarray,
    mu: Union[float, np.ndarray] = 1.0,
    omega: Union[float, np.ndarray] = 1.0,
    ell: Optional[Union[float, np.ndarray]] = None,
    dim: int = 2,
    max_num_iter: int = 100,
    eps: Optional[float] = None,
    x0: Optional[tuple] = None,
    isotropic: bool = False,
    verbose: Union[bool, int] = False,
    solver: Optional[Solver] = None,
    adaptive=None,
    promotion: Optional[str] = None,
    _context_cache: Optional[dict] = None,
) -> np.ndarray:
    """Split Bregman algorithm for TV denoising (split_bregman_tvd.py:14-177).

    Args: exactly those of ``darsia.split_bregman_tvd``.  ``x0 = (img0, d0, b0)``
    with ``d0``/``b0`` of shape ``(*img.shape, dim)``; as in the reference they are
    updated in place with the final split variables.

    ``promotion`` (extension, see the module docstring): ``"numpy2"`` -- the reference's own mixed flow for float32
    images, bit for bit; ``"native"`` -- arithmetic in the image dtype (2.5 x faster, close to the reference only while
    the shrinkage is inactive); ``None`` -- ``DXH_PROMOTION`` if set, else ``"numpy2"`` for every float32 image whose
    configuration has that flow (``numpy2_unsupported``), ``"native"`` otherwise.

    Returns:
        array: denoised image, same dtype as ``img``.
    """
    on_device = isinstance(img, DeviceArray)
    if not isinstance(img, np.ndarray) and not on_device:
        raise TypeError(f"Input type {type(img)} not supported.")
    if img.dtype.kind != "f":
        raise TypeError("split_bregman_tvd needs a float32/float64 image (integer images break the "
                        "reference's split variables, SURVEY.md quirk Q5)")
    if np.ndim(mu) == 0 and (ell is None or np.ndim(ell) == 0) and float(2 * mu if ell is None else ell) == 0.0:
        raise ZeroDivisionError("float division by zero")  # mu / ell of two Python floats (split_bregman_tvd.py:140,145)
    if solver is None:
        solver = Jacobi()
    if not isinstance(solver, (Jacobi, MG, CG)):
        raise NotImplementedError("inner solvers on the device path: darsia_amd.Jacobi, darsia_amd.MG, darsia_amd.CG")
    if isinstance(solver, MG) and solver.heterogeneous:
        raise ValueError("an MG solver constructed with coefficient arrays cannot be re-parametrised by the TV loop "
                         "(the reference's broadcasting fails on its coarse levels as well)")
    # the reference feeds the solver (split_bregman_tvd.py:62-68); kept for callers that inspect it
    # (ell = 2 mu of a device-resident mu is formed on the device; the solver object keeps its previous diffusion_coeff then)
    solver.update_params(mass_coeff=omega, dim=dim,
                         diffusion_coeff=ell if ell is not None else (None if isinstance(mu, DeviceArray) else 2 * mu))

    u0 = d0 = b0 = None
    if x0 is not None:
        u0, d0, b0 = x0
        for name, a in (("d0", d0), ("b0", b0)):
            if not isinstance(a, (np.ndarray, DeviceArray)) or tuple(a.shape) != tuple(img.shape) + (dim,):
                raise ValueError(f"{name} must have shape (*img.shape, dim) = {tuple(img.shape) + (dim,)}")

    quiet = not (verbose if isinstance(verbose, bool) else verbose > 0)
    why_not = numpy2_unsupported(img, mu, omega, ell, dim, x0, isotropic, solver, adaptive, quiet)
    if promotion in (None, "numpy2") and native_is_the_reference_flow(img, mu, omega, ell):
        promotion = "native"   # all-float32 arrays: the reference's own flow is float32 throughout
    promotion = choose_promotion(promotion, img.dtype, (mu, omega, ell), why_not)
    if promotion == "numpy2":
        if why_not is not None:
            raise NotImplementedError(f"promotion='numpy2': {why_not}")
        # ``_context_cache`` (a dict an operator object owns, restoration/tvd.py): the device context and its coefficient
        # planes -- a 256 MB float64 weight map at 4000 x 8000 -- stay from call to call while shape, device and the
        # coefficients' fingerprints are the same (in-place edits of a weight array: outside the contract, clear the dict)
        from ..concentration import _fingerprint

        if not quiet or x0 is not None or adaptive is not None:
            # verbose (one iteration per call so that increment and energy can be printed: the debug path, same messages as
            # split_bregman_tvd.py:103-104,161-167; the energy is evaluated in double from the device planes) and / or a
            # warm start (float32 d0 / b0 in, and back into the caller's arrays as the reference leaves them)
            show = (lambda it: verbose) if isinstance(verbose, bool) else (lambda it: verbose > 0 and it % verbose == 0)
            with DeviceTVD(img.shape, img.dtype, dim=dim, device=img.device if on_device else None, promotion="numpy2") as st:
                st.set_coeffs(mu, omega, ell, img)
                st.upload(img, u0, d0, b0)
                if adaptive is not None:
                    # the schedule is a host callable: one iteration per call, split variables explicit because the
                    # threshold changes between the iteration that writes d, b and the one that reads them
                    if not quiet:
                        print(f"The energy functional starts at {st.energy()}")
                    for it in range(max_num_iter):
                        r = st.run(1, solver.maxiter, solver.tol, eps=0.0 if eps is None else eps, flags=_lib.FLAG_EXPLICIT)
                        if adaptive(it):
                            st.adapt_ell()
                        if not quiet and show(it):
                            print(f"Split Bregman iteration {it} - relative increment: {r.rel_increment}, "
                                  f"energy functional: {st.energy(previous=True)}")
                        if eps is not None and r.converged:
                            break
                elif quiet:
                    st.run(max_num_iter, solver.maxiter, solver.tol, eps=eps)
                else:
                    print(f"The energy functional starts at {st.energy()}")
                    for it in range(max_num_iter):
                        r = st.run(1, solver.maxiter, solver.tol, eps=0.0 if eps is None else eps)
                        if show(it):
                            print(f"Split Bregman iteration {it} - relative increment: {r.rel_increment}, "
                                  f"energy functional: {st.energy(previous=True)}")
                        if eps is not None and r.converged:
                            break
                if x0 is None:
                    return st.download(out=DeviceArray(img.shape, img.dtype, img.device) if on_device else None)
                out_u = DeviceArray(img.shape, img.dtype, img.device) if on_device else None
                if isinstance(d0, DeviceArray) and isinstance(b0, DeviceArray):
                    u, _, _ = st.download(want_split=True, out=out_u, split_out=(d0, b0))   # the state stays in HBM
                    return u
                u, d, b = st.download(want_split=True, out=out_u)
                for dst, src in ((d0, d), (b0, b)):
                    if isinstance(dst, DeviceArray):
                        dst.set(src)
                    else:
                        dst[...] = src
                return u
        if (not on_device and eps is None and np.ndim(mu) == 0 and np.ndim(omega) == 0 and (ell is None or np.ndim(ell) == 0)
                and max_num_iter >= 1 and img.nbytes >= (32 << 20)
                and dim == 2 and solver.maxiter == 1 and solver.tol is None and (img.ndim == 2 or img.shape[2] <= 4)):
            # fixed iteration count on a large host image with scalar coefficients: row bands through the device, the
            # transfers overlapped with the iterations (bit-identical to the one-piece run below; array-valued mu / omega /
            # ell stay with the kept context, whose coefficient planes need not cross PCIe again)
            return solve_banded_numpy2(img, mu, omega, ell, max_num_iter)
        device = img.device if on_device else _lib.default_device()
        key = (tuple(img.shape), device, dim)
        mark = tuple(_fingerprint(v) for v in (mu, omega, ell))
        held = None if _context_cache is None else _context_cache.get("numpy2")
        if held is not None and held[0] != key:
            held[2].close()
            held = None
        if held is None:
            st = DeviceTVD(img.shape, img.dtype, dim=dim, device=device, promotion="numpy2")
            st_mark = None
        else:
            _, st_mark, st, _refs = held
        try:
            if st_mark != mark:
                st.set_coeffs(mu, omega, ell, img)
            st.upload(img)
            if on_device:
                st.run(max_num_iter, solver.maxiter, solver.tol, eps=eps)
                out = st.download(out=DeviceArray(img.shape, img.dtype, img.device))
            else:
                # the result array's pages are mapped on this thread WHILE the device iterates (the blocking C call runs
                # on a helper thread and holds no GIL): a fresh 384 MB array costs ~14 ms of page faults otherwise
                out = host_empty(img.shape, img.dtype)
                box = {}

                def work():
                    try:
                        st.run(max_num_iter, solver.maxiter, solver.tol, eps=eps)
                    except BaseException as e:  # noqa: BLE001 -- re-raised on the calling thread
                        box["e"] = e

                if out.nbytes >= (8 << 20):
                    th = threading.Thread(target=work)
                    th.start()
                    out.reshape(-1)[:: max(1, 4096 // out.itemsize)] = 0
                    th.join()
                else:
                    work()
                if "e" in box:
                    raise box["e"]
                out = st.download(out=out)
        except BaseException:
            st.close()
            if _context_cache is not None:
                _context_cache.pop("numpy2", None)
            raise
        if _context_cache is None:
            st.close()
        else:
            _context_cache["numpy2"] = (key, mark, st, (mu, omega, ell))
        return out
    if (not on_device and x0 is None and eps is None and adaptive is None and quiet and dim == 2 and img.ndim in (2, 3)
            and isinstance(solver, Jacobi) and solver.tol is None and solver.maxiter >= 1 and max_num_iter >= 1
            and img.nbytes >= (32 << 20)):
        # fixed iteration count on a large host image: row bands through the device, the transfers overlapped with the
        # iterations (dxh_tvd_solve_banded; bit-identical to the one-piece run below)
        return solve_banded(img, mu, omega, ell, max_num_iter, solver.maxiter, isotropic)

    with DeviceTVD(img.shape, img.dtype, dim=dim, device=img.device if on_device else None) as st:
        st.set_coeffs(mu, omega, ell, img)
        if not isinstance(solver, Jacobi):
            st.set_solver(solver)
        st.upload(img, u0, d0, b0)
        show = (lambda it: verbose) if isinstance(verbose, bool) else (lambda it: verbose > 0 and it % verbose == 0)
        if adaptive is not None:
            # the schedule is a host callable: one iteration per call, split variables kept explicit because
            # the shrinkage threshold changes between the iteration that writes d, b and the one that reads them
            for it in range(max_num_iter):
                r = st.run(1, solver.maxiter, solver.tol, eps=0.0 if eps is None else eps, isotropic=isotropic,
                           flags=_lib.FLAG_EXPLICIT)
                if adaptive(it):
                    st.adapt_ell()
                if eps is not None and r.converged:
                    break
        elif (verbose if isinstance(verbose, bool) else verbose > 0):
            # verbose: one iteration per call so that increment and energy can be printed (debug path,
            # same messages as split_bregman_tvd.py:103-104,161-167)
            print(f"The energy functional starts at {st.energy()}")
            for it in range(max_num_iter):
                r = st.run(1, solver.maxiter, solver.tol, eps=0.0 if eps is None else eps, isotropic=isotropic)
                if show(it):
                    print(f"Split Bregman iteration {it} - relative increment: {r.rel_increment}, "
                          f"energy functional: {st.energy(previous=True)}")
                if eps is not None and r.converged:
                    break
        else:
            # The result array is allocated and its pages are touched on this thread WHILE the device iterates (the
            # blocking C call runs on a helper thread and holds no GIL): a fresh 384 MB array costs ~14 ms of page
            # faults, which otherwise land inside the device-to-host copy (22-36 ms instead of 7 ms, measured).
            if on_device:
                # device in, device out: nothing crosses PCIe
                st.run(max_num_iter, solver.maxiter, solver.tol, eps=eps, isotropic=isotropic)
                if x0 is None:
                    return st.download(out=DeviceArray(img.shape, img.dtype, img.device))
            out = host_empty(img.shape, img.dtype) if (x0 is None and not on_device) else None
            box = {}

            def work():
                try:
                    box["r"] = st.run(max_num_iter, solver.maxiter, solver.tol, eps=eps, isotropic=isotropic)
                except BaseException as e:  # noqa: BLE001 -- re-raised on the calling thread
                    box["e"] = e

            if on_device:
                pass
            elif out is not None and out.nbytes >= (8 << 20):
                th = threading.Thread(target=work)
                th.start()
                out.reshape(-1)[:: max(1, 4096 // out.itemsize)] = 0
                th.join()
            else:
                work()
            if "e" in box:
                raise box["e"]
            if x0 is None:
                return st.download(out=out)
        if x0 is None:
            return st.download(out=DeviceArray(img.shape, img.dtype, img.device)) if on_device else st.download()
        u, d, b = st.download(want_split=True, out=DeviceArray(img.shape, img.dtype, img.device) if on_device else None)
        d0[...] = d
        b0[...] = b
        return u
