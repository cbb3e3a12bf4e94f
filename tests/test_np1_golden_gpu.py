"""The NATIVE float32 kernels (``promotion="native"`` -- the flow ``bench.py``'s headline times) against vectors
produced by EXECUTING the reference under NumPy 1.26 (tests/golden/make_golden_np1.py), where value-based casting
keeps the Jacobi iterate of a float32 image float32 (utils/linear_solvers/jacobi.py:101-114;
restoration/split_bregman_tvd.py:84-101,129-146).  Bar: BIT-EXACT -- the native flow IS the reference's own
arithmetic under NumPy < 2, incl. the regimes with an active shrinkage where it leaves the NumPy-2 flow by 10 - 80 x
rtol 1e-4 (``mu = 1``; ``ell = 1`` at ``mu = 0.1``)."""
import os
import sys

import numpy as np
import pytest

import darsia_amd as da

pytestmark = pytest.mark.gpu
NATIVE_DTYPE_FLOW = True   # conftest.py: this module pins the native float32 kernels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def np1():
    return np.load(os.path.join(ROOT, "tests", "golden", "tvd_np1_golden.npz"), allow_pickle=False)


def bits(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape, (a.dtype, b.dtype, a.shape, b.shape)
    if not np.array_equal(a, b):
        d = np.abs(a.astype(np.float64) - b)
        raise AssertionError(f"not bit-exact: {np.count_nonzero(d)} of {d.size} differ, max abs {d.max():.3e}")


CASES = [
    ("p2_aniso2d_f32", dict(mu=0.1, max_num_iter=20)),
    ("p3_aniso_rgb_f32", dict(mu=0.1, max_num_iter=10)),
    ("p3_aniso_rgb_odd_f32", dict(mu=0.15, omega=0.7, max_num_iter=7)),
    ("p3_aniso_2ch_odd_f32", dict(mu=0.1, max_num_iter=6)),
    ("p2_aniso2d_odd_f32", dict(mu=0.2, ell=0.3, max_num_iter=9)),
    ("p2_tiny_f32", dict(mu=0.1, max_num_iter=4)),
    ("p2_tiny_col_f32", dict(mu=0.1, max_num_iter=4)),
    ("p20_mu1_rgb_f32", dict(mu=1.0, max_num_iter=60)),
    ("p21_ell1_f32", dict(mu=0.1, ell=1.0, max_num_iter=100)),
    ("p22_mu1_gray_f32", dict(mu=1.0, omega=0.5, max_num_iter=40)),
    ("p9_vol3d_f32", dict(mu=0.1, dim=3, max_num_iter=6)),
]


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_native_flow_is_the_reference_under_numpy_1_bit_for_bit(np1, name, kw):
    img = np1[name + "/img"]
    assert img.dtype == np.float32 and str(np1["meta/numpy"]).startswith("1.")
    bits(da.split_bregman_tvd(img, promotion="native", **kw), np1[name + "/out"])
    # device-resident in / out, same bits
    d = da.split_bregman_tvd(da.DeviceArray.from_host(img), promotion="native", **kw)
    bits(d.get(), np1[name + "/out"])


@pytest.mark.parametrize("fuse_k", ["", "1", "2", "3"])
def test_every_launch_form_of_the_headline_agrees_with_the_reference(np1, monkeypatch, fuse_k):
    """The headline's K = 4 launches, the K = 1 replay its self-check uses, K = 2 / 3 and the three-kernel generic sweep:
    all the reference's bits."""
    from darsia_amd import _lib

    if fuse_k:
        monkeypatch.setenv("DXH_FUSE_K", fuse_k)
    img = np1["p20_mu1_rgb_f32/img"]
    for flags in (0, _lib.FLAG_NO_FUSED):
        with da.DeviceTVD(img.shape, img.dtype, promotion="native") as st:
            st.set_coeffs(1.0, 1.0, None, img)
            st.upload(img)
            st.run(60, flags=flags)
            bits(st.download(), np1["p20_mu1_rgb_f32/out"])


def test_several_sweeps(np1):
    img = np1["p6_jacobi3_rgb_f32/img"]
    bits(da.split_bregman_tvd(img, mu=0.1, max_num_iter=5, solver=da.Jacobi(maxiter=3), promotion="native"),
         np1["p6_jacobi3_rgb_f32/out"])


@pytest.mark.parametrize("name,eps", [("p7_eps_f32", 1e-3), ("p7_eps_rgb_f32", 2e-3)])
def test_early_stop(np1, name, eps):
    img = np1[name + "/img"]
    bits(da.split_bregman_tvd(img, mu=0.1, max_num_iter=100, eps=eps, promotion="native"), np1[name + "/out"])
    with da.DeviceTVD(img.shape, img.dtype, promotion="native") as st:
        st.set_coeffs(0.1, 1.0, None, img)
        st.upload(img)
        res = st.run(100, eps=eps)
        assert res.converged == 1 and res.iters_done == int(np1[name + "/iters"])


def test_float32_coefficient_arrays(np1):
    """Arrays are never value-cast: with float32 arrays the reference is float32 throughout under either NumPy; the
    DEFAULT call picks the native kernels for it (``native_is_the_reference_flow`` covers array omega / ell; an array mu
    next to scalar omega / ell is the numpy2 flow under NumPy >= 2 and this one under NumPy < 2)."""
    bits(da.split_bregman_tvd(np1["p5_het_mu_f32/img"], mu=np1["p5_het_mu_f32/mu"], omega=0.1, max_num_iter=12,
                              promotion="native"), np1["p5_het_mu_f32/out"])
    bits(da.split_bregman_tvd(np1["p5_het_mu_rgb_f32/img"], mu=np1["p5_het_mu_rgb_f32/mu"], max_num_iter=6,
                              promotion="native"), np1["p5_het_mu_rgb_f32/out"])


def test_warm_start_chain(np1):
    img = np1["p8_warm_f32/img"]
    d = np.zeros((*img.shape, 2), np.float32)
    b = np.zeros((*img.shape, 2), np.float32)
    first = da.split_bregman_tvd(img, mu=0.1, max_num_iter=5, x0=(img.copy(), d, b), promotion="native")
    bits(first, np1["p8_warm_f32/first"])
    bits(d, np1["p8_warm_f32/d5"])
    bits(b, np1["p8_warm_f32/b5"])
    second = da.split_bregman_tvd(img, mu=0.1, max_num_iter=5, x0=(first, d, b), promotion="native")
    bits(second, np1["p8_warm_f32/second"])
    bits(d, np1["p8_warm_f32/d10"])
    bits(b, np1["p8_warm_f32/b10"])


def test_config1_512_float32(np1):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import noisy_blocks  # pure NumPy helper (nothing of the reference is touched at import)
    big32 = noisy_blocks((512, 512), int(np1["p12_512/seed"]), np.float64).astype(np.float32)
    res = da.split_bregman_tvd(big32, mu=0.1, max_num_iter=20, promotion="native")
    bits(res[:16, :16], np1["p12_512/crop_tl_f32"])
    bits(res[248:264, 248:264], np1["p12_512/crop_mid_f32"])
    bits(res[-16:, -16:], np1["p12_512/crop_br_f32"])
    assert float(res.astype(np.float64).sum()) == float(np1["p12_512/sum_f32"])


def test_jacobi_and_h1_return_float32_like_the_reference_under_numpy_1(np1):
    img = np1["p10_jacobi4_f32/img"]
    s = da.Jacobi(maxiter=4)
    s.update_params(mass_coeff=1.3, diffusion_coeff=0.4, dim=2)
    bits(s(img.copy(), np.float32(1.3) * img, promotion="native"), np1["p10_jacobi4_f32/out"])
    rgb = np1["p10_h1_rgb_f32/img"]
    bits(da.H1_regularization(rgb, mu=0.4, omega=1.3, dim=2, solver=da.Jacobi(maxiter=4), promotion="native"),
         np1["p10_h1_rgb_f32/out"])
