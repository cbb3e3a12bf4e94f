"""The calls of the reference's example scripts, argument for argument, through this package (a user who switches keeps their
scripts): examples/regularization.py:8-47 and examples/paper/pet_simulations_comparison_block_b.py:680-713 -- volumes, a
heterogeneous weight, Jacobi(maxiter=20) / CG inside the TV loop and H1 regularisation, omega per voxel, verbose, eps."""
import contextlib
import io

import numpy as np
import pytest

import darsia_amd as darsia
from oracle import tvd_numpy as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def volume():
    rng = np.random.default_rng(0)
    vol = rng.random((24, 32, 48))
    return vol, darsia.ScalarImage(vol.copy(), dimensions=[0.1, 0.1, 0.1], space_dim=3)


def test_examples_regularization_calls(volume):
    vol, img = volume
    mu = np.ones_like(vol)
    mu[:, 0:12, :] = 0.5
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        res = darsia.tvd(img=img, method="heterogeneous bregman", isotropic=False, weight=mu, omega=0.1, max_num_iter=30, eps=1e-6,
                         dim=3, verbose=True, solver=darsia.Jacobi(maxiter=20))
    assert isinstance(res, darsia.ScalarImage) and res.img.shape == vol.shape and res.img.dtype == vol.dtype
    assert len(out.getvalue().splitlines()) >= 30          # the reference's per-iteration messages
    # (the TVD operator hands its `regularization`, 1.0 unless given, on as ell: restoration/tvd.py:37,104-112)
    ref = O.split_bregman_tvd(vol, mu=mu, omega=0.1, ell=1.0, max_num_iter=30, eps=1e-6, dim=3, jacobi_maxiter=20)
    assert np.array_equal(res.img, ref)
    h1 = darsia.H1_regularization(img=img, mu=1, omega=1, dim=3, solver=darsia.Jacobi(maxiter=30, tol=1e-6, verbose=True))
    assert np.array_equal(h1.img, O.h1_regularization(vol, 1, 1, dim=3, jacobi_maxiter=30, jacobi_tol=1e-6, promotion="strict"))


def test_pet_comparison_calls(volume):
    vol, img = volume
    omega = 0.01 + 0.02 * np.random.default_rng(1).random(vol.shape)
    h1 = darsia.H1_regularization(img, mu=0.1, omega=omega, dim=3, solver=darsia.CG(maxiter=10000, tol=1e-5))
    assert isinstance(h1, darsia.ScalarImage) and np.isfinite(h1.img).all()
    with contextlib.redirect_stdout(io.StringIO()):
        res = darsia.tvd(img, method="heterogeneous bregman", isotropic=True, weight=0.02, omega=omega, dim=3, max_num_iter=100,
                         eps=1e-5, verbose=True, solver=darsia.Jacobi(maxiter=20))
    ref = O.split_bregman_tvd(vol, mu=0.02, omega=omega, ell=1.0, isotropic=True, dim=3, max_num_iter=100, eps=1e-5, jacobi_maxiter=20)
    assert np.array_equal(res.img, ref)
