"""CPU restatement (NumPy) of DarSIA's split-Bregman TV denoising hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; ``darsia_amd`` never does (its operators fail loudly when the HIP
library is missing).

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against ``tests/golden/tvd_golden.npz``, which was produced by importing
the reference's own modules (``tests/golden/make_golden.py``).  The reference's
test-suite holds no numeric test for this path.

What is restated (reference file:line, relative to /root/reference/src/darsia):

* ``neighbor_sum``        utils/linear_solvers/jacobi.py:25-55
* ``jacobi``              utils/linear_solvers/jacobi.py:57-133 (without the
                          hidden ``const_diag`` cache, quirk Q3)
* ``forward_diff`` / ``backward_diff`` / ``laplace``
                          utils/derivatives.py:16-121
* ``split_bregman_tvd``   restoration/split_bregman_tvd.py:14-177
* ``h1_regularization``   restoration/h1_regularization.py:15-73

The formulation is this repo's own: planar per-axis split variables (lists of
arrays instead of a trailing interleaved axis), clamped-index gathers instead
of slice-accumulation, explicit ``dub`` state.  Floating-point operation ORDER
per element follows the reference so that results agree to the last bit in
float64 and float32.

dtype flow (``promotion``):

* ``"numpy2"`` -- what the reference does under NumPy >= 2 (NEP 50): with
  scalar coefficients its cached ``const_diag_scaled`` is a strongly typed
  ``np.float64`` (jacobi.py:104-106), so the Jacobi iterate of a float32 image
  is float64 while ``d``/``b``/``rhs`` stay float32.  Obtained here by building
  the coefficients with the same Python/NumPy scalar kinds and letting NumPy
  promote.
* ``"strict"`` -- every quantity in the image dtype (the NumPy 1.x flow, and
  the arithmetic the HIP kernels implement).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "neighbor_sum",
    "jacobi",
    "forward_diff",
    "backward_diff",
    "laplace",
    "shrink",
    "split_bregman_tvd",
    "h1_regularization",
]


def _take(x: np.ndarray, axis: int, idx: np.ndarray) -> np.ndarray:
    return np.take(x, idx, axis=axis)


def neighbor_sum(x: np.ndarray) -> np.ndarray:
    """Sum of the two axis-neighbours over EVERY array axis, replicate ghosts.

    jacobi.py:25-55.  Per element the reference accumulates, starting from 0,
    ``left, right`` for axis 0, then axis 1, ...; out-of-range neighbours are
    replaced by the element itself.  The same order is kept here.
    """
    acc = np.zeros_like(x)
    for ax in range(x.ndim):
        n = x.shape[ax]
        i = np.arange(n)
        acc = acc + _take(x, ax, np.maximum(i - 1, 0))
        acc = acc + _take(x, ax, np.minimum(i + 1, n - 1))
    return acc


def forward_diff(x: np.ndarray, axis: int, h=None) -> np.ndarray:
    """``x[i] - x[i-1]``, first entry ``x[0] - x[0]`` (derivatives.py:40-59)."""
    i = np.arange(x.shape[axis])
    r = x - _take(x, axis, np.maximum(i - 1, 0))
    return r if h is None else r / h


def backward_diff(x: np.ndarray, axis: int, h=None) -> np.ndarray:
    """``x[i+1] - x[i]``, last entry ``x[-1] - x[-1]`` (derivatives.py:16-37)."""
    n = x.shape[axis]
    i = np.arange(n)
    r = _take(x, axis, np.minimum(i + 1, n - 1)) - x
    return r if h is None else r / h


def laplace(x: np.ndarray, axis=None, dim: int = 2, h=None, diffusion_coeff=1) -> np.ndarray:
    """derivatives.py:62-121."""

    def one(ax):
        return 0.5 * (
            backward_diff(diffusion_coeff * forward_diff(x, ax, h), ax, h)
            + forward_diff(diffusion_coeff * backward_diff(x, ax, h), ax, h)
        )

    if axis is not None:
        return one(axis)
    res = np.zeros_like(x)
    for ax in range(dim):
        res += one(ax)
    return res


def shrink(x: np.ndarray, k) -> np.ndarray:
    """Soft threshold (split_bregman_tvd.py:80-81)."""
    return np.maximum(np.abs(x) - k, 0) * np.sign(x)


def _coefficients(omega, ell, dim, h, promotion, dtype):
    """Jacobi diagonal and scaled diagonal (jacobi.py:57-72, 101-106)."""
    diag = omega + ell * 2 * dim / h**2
    diag_scaled = np.divide(diag, ell / h**2)  # np.float64 when both are Python floats
    if promotion == "strict":
        if np.ndim(diag) == 0:
            diag = float(diag)
        else:
            diag = np.asarray(diag, dtype=dtype)
        if np.ndim(diag_scaled) == 0:
            diag_scaled = float(diag_scaled)  # weak scalar: no promotion
        else:
            diag_scaled = np.asarray(diag_scaled, dtype=dtype)
    return diag, diag_scaled


def jacobi(x0, rhs, mass_coeff, diffusion_coeff, dim=2, maxiter=1, tol=None, h=1.0,
           promotion="numpy2", coefficients=None):
    """``maxiter`` Jacobi sweeps for ``mass*x - diffusion*laplace(x) = rhs``.

    jacobi.py:74-133.  With ``tol`` the reference returns the iterate BEFORE
    the one that met the tolerance (quirk Q7); kept.
    """
    if coefficients is None:
        diag, diag_scaled = _coefficients(mass_coeff, diffusion_coeff, dim, h, promotion,
                                          np.asarray(x0).dtype)
    else:  # the reference's cached diagonal (quirk Q3): fixed at the solver's first call
        diag, diag_scaled = coefficients
    rhs_scaled = np.divide(rhs, diag)
    x = x0
    if tol is None:
        for _ in range(maxiter):
            x = rhs_scaled + np.divide(neighbor_sum(x), diag_scaled)
    else:
        for _ in range(maxiter):
            x_new = rhs_scaled + np.divide(neighbor_sum(x), diag_scaled)
            err = np.linalg.norm(x_new - x) / np.linalg.norm(x0)
            if err < tol:
                break
            x = x_new
    return x


def split_bregman_tvd(
    img: np.ndarray,
    mu=1.0,
    omega=1.0,
    ell=None,
    dim: int = 2,
    max_num_iter: int = 100,
    eps=None,
    x0=None,
    isotropic: bool = False,
    jacobi_maxiter: int = 1,
    jacobi_tol=None,
    promotion: str = "numpy2",
    return_state: bool = False,
    adaptive=None,
    solver=None,
):
    """split_bregman_tvd.py:14-177 with a fresh Jacobi solver, or with ``solver`` (an object with
    ``update_params(mass_coeff=, diffusion_coeff=, dim=)`` and ``__call__(x0, rhs)``, e.g.
    oracle.solvers_numpy.MG / CG) in the place of the Jacobi sweeps (split_bregman_tvd.py:63-67,133).

    ``adaptive(iteration) -> bool`` (split_bregman_tvd.py:149-158): when true, ``ell`` becomes the per-pixel
    ``1 / max(|grad u|_1, 1e-12)``.  As in the reference, the Jacobi solver keeps the diagonal it cached at
    its first call (quirk Q3); only the right-hand side and the shrinkage threshold follow the new ``ell``.

    ``x0 = (u0, [d_0..d_{dim-1}], [b_0..b_{dim-1}])`` uses PLANAR split
    variables (one array per axis).  With ``return_state`` the result is
    ``(u, d_list, b_list, iterations_done)``.
    """
    assert promotion in ("numpy2", "strict")
    dt = img.dtype
    assert dt.kind == "f", "float-only contract (SURVEY.md quirk Q5)"
    if promotion == "strict":
        cast = lambda c: c if np.ndim(c) == 0 else np.asarray(c, dtype=dt)  # noqa: E731
        mu, omega = cast(mu), cast(omega)
        if ell is not None:
            ell = cast(ell)
    img_nrm = np.linalg.norm(img)
    if ell is None:
        ell = 2 * mu
    if x0 is None:
        u = img.copy()
        d = [np.zeros(img.shape, dt) for _ in range(dim)]
        b = [np.zeros(img.shape, dt) for _ in range(dim)]
    else:
        u, d, b = x0
        d = [np.array(a, dtype=dt) for a in d]
        b = [np.array(a, dtype=dt) for a in b]

    done = 0
    jac_coef = _coefficients(omega, ell, dim, 1.0, promotion, dt)  # cached by the reference's solver
    if solver is not None:
        solver.update_params(mass_coeff=omega, diffusion_coeff=ell, dim=dim)
    for it in range(max_num_iter):
        rhs = np.multiply(omega, img)
        for j in range(dim):
            # the reference accumulates IN PLACE (``result += ...``, split_bregman_tvd.py:84-89): the sum keeps the dtype
            # of omega * img even when ell -- hence the difference -- is a float64 array next to a float32 image
            rhs = (rhs + forward_diff(ell * (b[j] - d[j]), j)).astype(rhs.dtype, copy=False)
        if solver is not None:
            u_new = solver(u, rhs)
        else:
            u_new = jacobi(u, rhs, omega, ell, dim=dim, maxiter=jacobi_maxiter, tol=jacobi_tol,
                           promotion=promotion, coefficients=jac_coef)
        if promotion == "strict":
            u_new = u_new.astype(dt, copy=False)
        q = mu / ell
        if isotropic:
            dub = [(b[j] + backward_diff(u_new, j)).astype(dt) for j in range(dim)]
            s = np.zeros(img.shape, dt)
            for j in range(dim):
                s = s + dub[j] ** 2
            s = np.sqrt(s)
            fac = np.maximum(s - q, 0) / (s + 1e-18)
            for j in range(dim):
                d[j] = (dub[j] * fac).astype(dt)
                b[j] = (dub[j] * (1 - fac)).astype(dt)
        else:
            for j in range(dim):
                dub = backward_diff(u_new, j) + b[j]
                d[j] = shrink(dub, q).astype(dt)
                b[j] = (dub - d[j]).astype(dt)
        if adaptive is not None and adaptive(it):
            l1 = np.zeros(img.shape, dt)
            for j in range(dim):
                l1 = l1 + np.abs(backward_diff(u_new, j).astype(dt))
            ell = 1.0 / np.maximum(l1, 1e-12)
        rel = np.linalg.norm(u_new - u) / img_nrm
        u = u_new
        done = it + 1
        if eps is not None and rel < eps:
            break
    out = u.astype(dt, copy=False)
    if return_state:
        return out, d, b, done
    return out


def h1_regularization(img, mu, omega=1.0, dim=2, jacobi_maxiter=1, jacobi_tol=None,
                      promotion="numpy2"):
    """h1_regularization.py:15-73: one solver call on ``omega * img``; arrays with
    more than ``dim`` axes are processed slice by slice over the trailing axes."""
    dt = img.dtype

    def solve(a):
        return jacobi(a, omega * a, omega, mu, dim=dim, maxiter=jacobi_maxiter,
                      tol=jacobi_tol, promotion=promotion)

    if img.ndim != dim:
        moved = np.moveaxis(img, list(range(dim)), list(range(-dim, 0)))
        res = np.zeros_like(moved)
        for idx in np.ndindex(moved.shape[:-dim]):
            res[idx] = solve(moved[idx])
        out = np.moveaxis(res, list(range(-dim, 0)), list(range(dim)))
    else:
        out = solve(img)
    return out.astype(dt, copy=False)
