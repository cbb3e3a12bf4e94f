"""H1 regularisation (reference: restoration/h1_regularization.py:15-146): one
Jacobi solve of ``omega*u - mu*laplace(u) = omega*img``; arrays with more axes
than ``dim`` are treated channel by channel (no channel coupling here, unlike
the TV path -- h1_regularization.py:54-67)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .. import _lib
from ..device import DeviceArray, host_empty
from ..image import with_array
from ..utils.linear_solvers.jacobi import Jacobi, float32_flow
from ..utils.linear_solvers.solver import Solver


def _h1_device(img: DeviceArray, mu: float, omega: float, dim: int, solver: Solver, promotion=None) -> DeviceArray:
    """Device-resident image in, device-resident result out (the reference chains in host memory,
    h1_regularization.py:39-73): the Jacobi solve with rhs = omega * img formed on the device; MG / CG objects get the
    right-hand side as a second device array.  Channel-wise (``img.ndim == dim + 1``) only in the one-launch form, i.e.
    Jacobi without a tolerance."""
    if type(omega) not in (float, int):
        raise NotImplementedError("device-resident H1_regularization needs omega as a Python number")
    if not isinstance(solver, Jacobi):
        if img.ndim != dim:
            raise NotImplementedError("device-resident H1_regularization with MG / CG: arrays of exactly dim axes")
        rhs = DeviceArray(img.shape, img.dtype, img.device)
        try:
            _lib.check(_lib.load().dxh_scale(img.device, _lib.dtype_code(img.dtype), img.size, img.ptr, float(omega), rhs.ptr))
            return solver(img, rhs)
        finally:
            rhs.close()
    if img.ndim == dim + 1 and dim == 2 and solver.tol is None:
        nsum = _lib.NSUM_SPATIAL
    elif img.ndim == dim:
        nsum = _lib.NSUM_ALL_AXES
    else:
        raise NotImplementedError("device-resident H1_regularization: (H, W[, C]) with Jacobi(tol=None), or arrays of dim axes")
    if img.dtype == np.float32 and float32_flow(promotion, omega, mu) == "numpy2":
        # the reference's own float32 flow; convert_dtype(out, float32) on the way out (h1_regularization.py:76)
        return solver._call_numpy2(img, None, 1.0, out_dtype=np.float32, nsum=nsum)
    ndim, n0, n1, n2 = _lib.shape3(img.shape, dim)
    out = DeviceArray(img.shape, img.dtype, img.device)
    done = C.c_int(0)
    _lib.check(_lib.load().dxh_jacobi_solve(
        img.device, _lib.dtype_code(img.dtype), ndim, dim, nsum, n0, n1, n2, img.ptr, None, out.ptr,
        float(omega), None, float(mu), None, 1.0, int(solver.maxiter), -1.0 if solver.tol is None else float(solver.tol),
        C.byref(done)))
    return out


def _h1_array(img: np.ndarray, mu: float, omega: float, dim: int, solver: Solver, promotion=None) -> np.ndarray:
    if img.dtype.kind != "f":
        raise TypeError("H1_regularization needs a float32/float64 image")
    solver.update_params(mass_coeff=omega, diffusion_coeff=mu, dim=dim)
    arrays = any(isinstance(c, DeviceArray) or np.ndim(c) != 0 for c in (mu, omega))
    if isinstance(img, DeviceArray):
        if arrays:
            raise NotImplementedError("device-resident H1_regularization: scalar mu and omega (array-valued coefficients: pass a "
                                      "host image, or call the solver on device arrays yourself)")
        return _h1_device(img, mu, omega, dim, solver, promotion)
    if not isinstance(solver, Jacobi) or arrays:
        # MG / CG objects (device solvers themselves), and Jacobi with array-valued mu / omega (the reference's own volume
        # workload: examples/paper/pet_simulations_comparison_block_b.py:680-686 passes omega per voxel): the reference's
        # generic flow, channel by channel (h1_regularization.py:48-76) -- the solver object does the work on the device
        if not callable(solver):
            raise TypeError("solver must be a darsia_amd Solver object")
        if img.ndim != dim:
            moved = np.moveaxis(img, np.arange(dim).tolist(), np.arange(-dim, 0).tolist())
            res = np.zeros_like(moved)
            for idx in np.ndindex(moved.shape[:-dim]):
                plane = np.ascontiguousarray(moved[idx])
                res[idx] = solver(plane, omega * plane)
            return np.ascontiguousarray(np.moveaxis(res, np.arange(-dim, 0).tolist(), np.arange(dim).tolist()))
        out = solver(img, omega * img)
        # convert_dtype(out_img, img_dtype) (h1_regularization.py:76): Jacobi hands a float64 array back for float32 operands
        return out if out.dtype == img.dtype else out.astype(img.dtype)
    code = _lib.dtype_code(img.dtype)
    _lib.require_gpu()
    src = np.ascontiguousarray(img)
    if img.ndim == dim + 1 and solver.tol is not None:
        # with a tolerance every channel stops at its own sweep and returns its own iterate (the reference solves channel
        # by channel, h1_regularization.py:54-67; Jacobi's stop test is a norm over the array it is given, jacobi.py:116-126)
        moved = np.moveaxis(src, -1, 0)
        res = np.stack([_h1_array(np.ascontiguousarray(ch), mu, omega, dim, solver, promotion) for ch in moved], axis=0)
        return np.ascontiguousarray(np.moveaxis(res, 0, -1))
    if img.ndim == dim + 1 and dim == 2:
        # all channels in one launch: neighbour sum restricted to the spatial axes
        ndim, n0, n1, n2 = _lib.shape3(img.shape, dim)
        nsum = _lib.NSUM_SPATIAL
    elif img.ndim == dim:
        ndim, n0, n1, n2 = _lib.shape3(img.shape, dim)
        nsum = _lib.NSUM_ALL_AXES
    else:
        raise ValueError(f"image with {img.ndim} axes and dim={dim} not supported")
    if img.dtype == np.float32 and type(omega) in (float, int) and float32_flow(promotion, omega, mu) == "numpy2":
        # the reference's own float32 flow (float64 iterate from the first sweep's division on); the result rounded to
        # float32 as convert_dtype(out_img, float32) does (h1_regularization.py:76)
        return solver._call_numpy2(src, None, 1.0, out_dtype=np.float32, nsum=nsum)
    # omega * img: on the device when omega is a Python number (NumPy rounds it to the image dtype and multiplies there:
    # the kernel does the same); any other scalar type goes through NumPy's own promotion on the host
    rhs = None if type(omega) in (float, int) else np.ascontiguousarray(omega * src, dtype=img.dtype)
    out = host_empty(src.shape, src.dtype)
    done = C.c_int(0)
    _lib.check(_lib.load().dxh_jacobi_solve(
        _lib.default_device(), code, ndim, dim, nsum, n0, n1, n2, _lib.ptr(src), _lib.ptr(rhs), _lib.ptr(out),
        float(omega), None, float(mu), None, 1.0, int(solver.maxiter), -1.0 if solver.tol is None else float(solver.tol),
        C.byref(done)))
    if getattr(solver, "verbose", False):   # (what Jacobi.__call__ prints: the reference's line per sweep, jacobi.py:115,128)
        from ..utils.linear_solvers.jacobi import _verbose_lines
        _verbose_lines(done.value, solver.maxiter, solver.tol)
    return out


def H1_regularization(img, mu: float, omega: float = 1.0, dim: int = 2, solver: Optional[Solver] = None, promotion=None):
    """Inline application of H1 regularisation to an array or an Image-like object.  ``promotion`` (extension): the dtype
    flow of a float32 image with the Jacobi solver -- ``None``: ``DXH_PROMOTION``, else the reference's own ("numpy2":
    float64 iterate, float32 result, utils/linear_solvers/jacobi.py); ``"native"``: float32 throughout."""
    if solver is None:
        solver = Jacobi()
    if isinstance(img, (np.ndarray, DeviceArray)):
        return _h1_array(img, mu, omega, dim, solver, promotion)
    if hasattr(img, "img") and hasattr(img, "copy"):
        return with_array(img, _h1_array(img.img, mu, omega, dim, solver, promotion))
    raise TypeError(f"Type {type(img)} not supported.")
