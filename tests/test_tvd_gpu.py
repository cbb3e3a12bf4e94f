"""Parity of the HIP path (through the C ABI) with the oracle and the golden
vectors.  Needs a real MI355X: ``pytest -m gpu``.

Bars (north star: rtol 1e-4 vs the CPU path):
* vs the oracle's same-dtype flow ("strict"): BIT-EXACT -- the kernels perform
  the same IEEE operations in the same order (-ffp-contract=off);
* vs the golden vectors from the imported reference: float64 bit-exact;
  float32 within rtol 1e-4 / atol 1e-6 (the reference promotes the float32
  Jacobi iterate to float64 under NumPy >= 2, the kernels stay in float32).
"""
import numpy as np
import pytest

import darsia_amd as da
from darsia_amd import _lib
from oracle import tvd_numpy as O
from conftest import split_last_axis

pytestmark = pytest.mark.gpu
NATIVE_DTYPE_FLOW = True   # conftest.py: this module pins the native float32 kernels (DXH_PROMOTION=native)

RTOL, ATOL = 1e-4, 1e-6

VARIANTS = {"fused_dpp": 0, "fused_bpermute": _lib.FLAG_NO_DPP, "three_kernel": _lib.FLAG_NO_FUSED}


def run_device(img, mu=1.0, omega=1.0, ell=None, dim=2, n_iter=10, eps=None, isotropic=False, jm=1, jtol=None,
               flags=0, x0=None, want_split=False):
    with da.DeviceTVD(img.shape, img.dtype, dim=dim) as st:
        st.set_coeffs(mu, omega, ell, img)
        if x0 is None:
            st.upload(img)
        else:
            st.upload(img, *x0)
        res = st.run(n_iter, jm, jtol, eps=eps, isotropic=isotropic, flags=flags)
        out = st.download(want_split=want_split)
    return out, res


def assert_bits(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape
    if not np.array_equal(a, b):
        diff = np.abs(a.astype(np.float64) - b.astype(np.float64))
        raise AssertionError(f"not bit-exact: {np.count_nonzero(diff)} of {diff.size} differ, max abs {diff.max():.3e}")


CASES = [
    ("g1_aniso2d_f64", dict(mu=0.1), 20),
    ("g2_aniso2d_f32", dict(mu=0.1), 20),
    ("g3_aniso_rgb_f32", dict(mu=0.1), 10),
    ("g3_aniso_rgb_f64", dict(mu=0.1), 10),
    ("g3_aniso_rgb_odd_f32", dict(mu=0.15, omega=0.7), 7),
    ("g3_aniso_2ch_odd_f32", dict(mu=0.1), 6),
    ("g2_aniso2d_odd_f32", dict(mu=0.2, ell=0.3), 9),
    ("g2_tiny_f32", dict(mu=0.1), 4),
    ("g2_tiny_col_f32", dict(mu=0.1), 4),
    ("g2_single_px_f32", dict(mu=0.1), 3),
]


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("name,kw,n", CASES, ids=[c[0] for c in CASES])
def test_anisotropic_scalar(golden, name, kw, n, variant):
    img = golden[name + "/img"]
    out, res = run_device(img, n_iter=n, flags=VARIANTS[variant], **kw)
    assert res.iters_done == n
    assert_bits(out, O.split_bregman_tvd(img, max_num_iter=n, promotion="strict", **kw))
    ref = golden[name + "/out"]
    if img.dtype == np.float64:
        assert_bits(out, ref)
    else:
        np.testing.assert_allclose(out, ref, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_isotropic(golden, variant):
    img = golden["g4_iso2d_f64/img"]
    out, _ = run_device(img, mu=0.1, n_iter=10, isotropic=True, flags=VARIANTS[variant])
    assert_bits(out, O.split_bregman_tvd(img, mu=0.1, max_num_iter=10, isotropic=True, promotion="strict"))
    np.testing.assert_allclose(out, golden["g4_iso2d_f64/out"], rtol=1e-13, atol=0)
    # extension beyond the reference: float32 and colour
    rgb = golden["g3_aniso_rgb_f32/img"]
    out, _ = run_device(rgb, mu=0.1, n_iter=6, isotropic=True, flags=VARIANTS[variant])
    assert_bits(out, O.split_bregman_tvd(rgb, mu=0.1, max_num_iter=6, isotropic=True, promotion="strict"))


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_heterogeneous_coefficients(golden, variant):
    g, fl = golden, VARIANTS[variant]
    for name, kw, n in (
        ("g5_het_mu_f64", dict(mu=g["g5_het_mu_f64/mu"], omega=0.1), 12),
        ("g5_het_mu_f32", dict(mu=g["g5_het_mu_f32/mu"], omega=0.1), 12),
        ("g5_het_omega_f64", dict(mu=0.1, omega=g["g5_het_omega_f64/omega"], ell=0.25), 8),
        ("g5_het_mu_rgb_f32", dict(mu=g["g5_het_mu_rgb_f32/mu"]), 6),
    ):
        img = g[name + "/img"]
        out, _ = run_device(img, n_iter=n, flags=fl, **kw)
        assert_bits(out, O.split_bregman_tvd(img, max_num_iter=n, promotion="strict", **kw))
        if img.dtype == np.float64:
            assert_bits(out, g[name + "/out"])
        else:
            np.testing.assert_allclose(out, g[name + "/out"], rtol=RTOL, atol=ATOL)


def test_inner_solver_variants(golden):
    g = golden
    out, _ = run_device(g["g6_jacobi3_f64/img"], mu=0.1, n_iter=8, jm=3)
    assert_bits(out, g["g6_jacobi3_f64/out"])
    out, _ = run_device(g["g6_jacobi3_rgb_f32/img"], mu=0.1, n_iter=5, jm=3)
    assert_bits(out, O.split_bregman_tvd(g["g6_jacobi3_rgb_f32/img"], mu=0.1, max_num_iter=5, jacobi_maxiter=3,
                                         promotion="strict"))
    np.testing.assert_allclose(out, g["g6_jacobi3_rgb_f32/out"], rtol=RTOL, atol=ATOL)
    out, _ = run_device(g["g6_jacobi_tol_f64/img"], mu=0.1, n_iter=6, jm=20, jtol=1e-3)
    assert_bits(out, g["g6_jacobi_tol_f64/out"])


@pytest.mark.parametrize("name,eps", [("g7_eps_f64", 1e-3), ("g7_eps_f32", 1e-3), ("g7_eps_rgb_f32", 2e-3)])
def test_early_stop(golden, name, eps):
    img = golden[name + "/img"]
    out, res = run_device(img, mu=0.1, n_iter=100, eps=eps)
    assert res.converged == 1
    assert res.iters_done == int(golden[name + "/iters"])
    np.testing.assert_allclose(out, golden[name + "/out"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_warm_start_in_reference_layout(golden, tag):
    """x0 = (img0, d0, b0) with (*shape, dim) split variables, updated in place (split_bregman_tvd.py:93-97)."""
    g = golden
    img = g[f"g8_warm_{tag}/img"]
    d = np.zeros(img.shape + (2,), img.dtype)
    b = np.zeros(img.shape + (2,), img.dtype)
    first = da.split_bregman_tvd(img, mu=0.1, max_num_iter=5, x0=(img.copy(), d, b))
    tol = dict(rtol=0, atol=0) if tag == "f64" else dict(rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(first, g[f"g8_warm_{tag}/first"], **tol)
    np.testing.assert_allclose(d, g[f"g8_warm_{tag}/d5"], **tol)
    np.testing.assert_allclose(b, g[f"g8_warm_{tag}/b5"], **tol)
    second = da.split_bregman_tvd(img, mu=0.1, max_num_iter=5, x0=(first, d, b))
    np.testing.assert_allclose(second, g[f"g8_warm_{tag}/second"], **tol)
    np.testing.assert_allclose(d, g[f"g8_warm_{tag}/d10"], **tol)
    np.testing.assert_allclose(b, g[f"g8_warm_{tag}/b10"], **tol)
    # and 5 + 5 warm-started iterations equal 10 straight ones bit for bit
    straight = da.split_bregman_tvd(img, mu=0.1, max_num_iter=10)
    assert_bits(second, straight)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_volume_dim3(golden, tag):
    img = golden[f"g9_vol3d_{tag}/img"]
    out = da.split_bregman_tvd(img, mu=0.1, dim=3, max_num_iter=6)
    assert_bits(out, O.split_bregman_tvd(img, mu=0.1, dim=3, max_num_iter=6, promotion="strict"))
    if tag == "f64":
        assert_bits(out, golden[f"g9_vol3d_{tag}/out"])
    else:
        np.testing.assert_allclose(out, golden[f"g9_vol3d_{tag}/out"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("tag", ["2d_f64", "2d_f32", "rgb_f32"])
def test_h1_regularization(golden, tag):
    img = golden[f"g10_h1_{tag}/img"]
    out = da.H1_regularization(img, mu=0.4, omega=1.3, dim=2, solver=da.Jacobi(maxiter=4))
    assert_bits(out, O.h1_regularization(img, mu=0.4, omega=1.3, dim=2, jacobi_maxiter=4, promotion="strict"))
    if img.dtype == np.float64:
        assert_bits(out, golden[f"g10_h1_{tag}/out"])
    else:
        np.testing.assert_allclose(out, golden[f"g10_h1_{tag}/out"], rtol=RTOL, atol=ATOL)


def test_jacobi_operator(golden):
    img = golden["g1_aniso2d_f64/img"]
    s = da.Jacobi(maxiter=5, mass_coeff=1.0, diffusion_coeff=0.7, dim=2)
    assert_bits(s(img, 1.0 * img), O.jacobi(img, 1.0 * img, 1.0, 0.7, dim=2, maxiter=5))
    s = da.Jacobi(maxiter=30, tol=1e-2, mass_coeff=1.0, diffusion_coeff=0.7, dim=2)
    assert_bits(s(img, 1.0 * img), O.jacobi(img, 1.0 * img, 1.0, 0.7, dim=2, maxiter=30, tol=1e-2))


def test_derivatives(golden):
    g = golden
    a2, a3 = g["g11_diff/a2"], g["g11_diff/a3"]
    for ax in range(2):
        assert_bits(da.backward_diff(a2, ax), g[f"g11_diff/a2_bwd{ax}"])
        assert_bits(da.forward_diff(a2, ax), g[f"g11_diff/a2_fwd{ax}"])
        assert_bits(da.backward_diff(a2, ax, h=0.5), g[f"g11_diff/a2_bwd{ax}_h"])
        assert_bits(da.backward_diff(a3, ax), g[f"g11_diff/a3_bwd{ax}"])
        assert_bits(da.forward_diff(a3, ax), g[f"g11_diff/a3_fwd{ax}"])
    assert_bits(da.backward_diff(a3, 2, dim=3), g["g11_diff/a3_bwd2"])
    assert_bits(da.forward_diff(a3, 2, dim=3), g["g11_diff/a3_fwd2"])
    assert_bits(da.laplace(a2), g["g11_diff/a2_lap"])
    assert_bits(da.laplace(a2, h=0.5, diffusion_coeff=2.0), g["g11_diff/a2_lap_h"])
    assert_bits(da.laplace(a3, dim=3), g["g11_diff/a3_lap3"])
    assert_bits(da.laplace(a2, axis=1), g["g11_diff/a2_lap_ax1"])


def test_tvd_operator_on_array_and_image(golden):
    img = golden["g13_tvd_call/img"]
    op = da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=6, eps=None)
    out = op(img)
    np.testing.assert_allclose(out, golden["g13_tvd_call/out"], rtol=RTOL, atol=ATOL)
    im = da.Image(img.copy(), tag="frame")
    res = op(im)
    assert isinstance(res, da.Image) and res is not im and res.metadata()["tag"] == "frame"
    assert_bits(res.img, out)
    assert_bits(im.img, img)  # input untouched
    assert_bits(da.tvd(img, method="heterogeneous bregman", weight=0.1, max_num_iter=6, eps=None), out)


def test_config1_512_through_public_api(golden):
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import noisy_blocks
    g = golden
    big = noisy_blocks((512, 512), int(g["g12_512/seed"]), np.float64)
    res = da.split_bregman_tvd(big, mu=0.1, max_num_iter=20)
    assert_bits(res[:16, :16], g["g12_512/crop_tl"])
    assert_bits(res[-16:, -16:], g["g12_512/crop_br"])
    assert_bits(res[248:264, 248:264], g["g12_512/crop_mid"])
    assert res.sum() == float(g["g12_512/sum"])
    res32 = da.split_bregman_tvd(big.astype(np.float32), mu=0.1, max_num_iter=20)
    np.testing.assert_allclose(res32[248:264, 248:264], g["g12_512/crop_mid_f32"], rtol=RTOL, atol=ATOL)


def test_wide_image_many_strips_and_chunks():
    """Sizes that exercise several column strips, row chunks and ragged edges of the fused sweep."""
    rng = np.random.default_rng(5)
    for shape in ((70, 523, 3), (301, 260), (9, 1031, 4), (64, 249, 3), (130, 245, 3)):
        img = rng.random(shape).astype(np.float32)
        ref = O.split_bregman_tvd(img, mu=0.1, max_num_iter=4, promotion="strict")
        for fl in VARIANTS.values():
            out, _ = run_device(img, mu=0.1, n_iter=4, flags=fl)
            assert_bits(out, ref)


@pytest.mark.parametrize("fuse_k", [1, 2, 3, 4])
def test_temporal_blocking_is_bit_exact(monkeypatch, fuse_k):
    """K outer iterations per launch (k_fused_multi) must equal K single launches / the oracle bit for bit,
    for iteration counts that are and are not multiples of K, tall images (several row chunks, first/last
    chunk at the image border), tiny images (fewer rows than pipeline stages) and both dtypes."""
    monkeypatch.setenv("DXH_FUSE_K", str(fuse_k))
    rng = np.random.default_rng(17)
    cases = [((70, 523, 3), np.float32, 7), ((1500, 230, 3), np.float32, 6), ((3, 40, 3), np.float32, 5),
             ((1, 17), np.float32, 4), ((2, 300), np.float32, 9), ((257, 300), np.float64, 5),
             ((90, 130, 3), np.float64, 8), ((640, 700), np.float32, 12)]
    for shape, dt, n in cases:
        img = rng.random(shape).astype(dt)
        ref = O.split_bregman_tvd(img, mu=0.1, max_num_iter=n, promotion="strict")
        out, res = run_device(img, mu=0.1, n_iter=n)
        assert res.iters_done == n
        assert_bits(out, ref)
    img = rng.random((120, 260, 3)).astype(np.float32)
    out, _ = run_device(img, mu=0.1, n_iter=6, isotropic=True)
    assert_bits(out, O.split_bregman_tvd(img, mu=0.1, max_num_iter=6, isotropic=True, promotion="strict"))


# ---- Chambolle (default TVD() method) --------------------------------------------------------------
import os as _os

from oracle import chambolle_numpy as OC


@pytest.fixture(scope="module")
def sk():
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    return np.load(_os.path.join(root, "tests", "golden", "skimage_0_18_3_golden.npz"))


@pytest.mark.parametrize("name", ["c2d_f64", "c2d_f32", "c2d_fixed_f64", "c2d_fixed_f32", "crgb_f32",
                                  "crgb_fixed_f64", "c2d_one_iter", "c2d_tiny"])
def test_chambolle_vs_skimage_golden(sk, name):
    """Bit-exact per iteration; the stop iteration may differ only if the energy test is within rounding
    (energies are reduced in double here, in the image dtype with pairwise sums in NumPy)."""
    img = sk[name + "/img"]
    w, eps, n = (float(v) for v in sk[name + "/kw"])
    out, iters = da.denoise_tv_chambolle(img, weight=w, eps=eps, max_num_iter=int(n), return_iters=True)
    ref, ref_iters = OC.denoise_tv_chambolle(img, weight=w, eps=eps, max_num_iter=int(n), return_iters=True)
    assert iters == ref_iters
    assert_bits(out, ref)
    assert_bits(out, sk[name + "/out"])


def test_default_tvd_is_chambolle(sk):
    img = sk["crgb_f32/img"]
    out = da.TVD()(img)  # method "chambolle", weight 0.1, max_num_iter 200, eps 2e-4 (tvd.py:31-43)
    assert_bits(out, sk["crgb_f32/out"])
    assert_bits(da.tvd(img, weight=0.1), out)


@pytest.mark.parametrize("fuse_k", [1, 2, 3, 4])
@pytest.mark.parametrize("name,eps", [("g7_eps_f64", 1e-3), ("g7_eps_f32", 1e-3), ("g7_eps_rgb_f32", 2e-3)])
def test_early_stop_with_temporal_blocking(golden, monkeypatch, name, eps, fuse_k):
    """The eps test runs on K-iteration launches (increments of every fused iteration are reduced inside the
    kernel); a hit at an inner iteration is redone with exactly that many iterations.  Same stop iteration
    and same result as the per-iteration path / the reference."""
    monkeypatch.setenv("DXH_FUSE_K", str(fuse_k))
    img = golden[name + "/img"]
    out, res = run_device(img, mu=0.1, n_iter=100, eps=eps)
    assert res.converged == 1 and res.iters_done == int(golden[name + "/iters"])
    ref = O.split_bregman_tvd(img, mu=0.1, max_num_iter=res.iters_done, promotion="strict")
    assert_bits(out, ref)
    # eps never met: all iterations run
    out2, res2 = run_device(img, mu=0.1, n_iter=7, eps=1e-12)
    assert res2.converged == 0 and res2.iters_done == 7
    assert_bits(out2, O.split_bregman_tvd(img, mu=0.1, max_num_iter=7, promotion="strict"))


@pytest.mark.parametrize("m2", ["64", "68", "128", "184", "256", "16"])
@pytest.mark.parametrize("kind", ["aniso", "iso", "het"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_vectorised_volume_march_every_patch_shape(monkeypatch, m2, kind, dtype):
    """k_march3d_v (volumes whose rows are 16-byte multiples: four adjacent float / two adjacent double columns per thread) with each of its patch shapes forced
    (patch widths from 16 to 256 columns, incl. 68 and 184 whose thread rows do not divide 256), on volumes that span several patches along both in-plane axes with ragged ends, against the
    three kernels and the oracle; and with several plane chunks (DXH_NCHUNKS)."""
    monkeypatch.setenv("DXH_MARCH_M2", m2)
    rng = np.random.default_rng(int(m2))
    for shape, chunks in (((11, 21, 268), None), ((26, 37, 132), "5"), ((2, 3, 4), None)):
        if chunks:
            monkeypatch.setenv("DXH_NCHUNKS", chunks)
        else:
            monkeypatch.delenv("DXH_NCHUNKS", raising=False)
        vol = rng.random(shape).astype(dtype)
        mu = (0.03 + 0.2 * rng.random(shape)).astype(dtype) if kind == "het" else 0.1
        outs = {}
        for name, flags in (("march", 0), ("three_kernel", _lib.FLAG_NO_FUSED)):
            with da.DeviceTVD(vol.shape, vol.dtype, dim=3) as st:
                st.set_coeffs(mu, 0.7, None, vol)
                st.upload(vol)
                st.run(4, isotropic=kind == "iso", flags=flags)
                outs[name] = st.download(want_split=True)
        for a, b in zip(outs["march"], outs["three_kernel"]):
            assert_bits(a, b)
        assert_bits(outs["march"][0], O.split_bregman_tvd(vol, mu=mu, omega=0.7, dim=3, max_num_iter=4, isotropic=kind == "iso",
                                                          promotion="strict"))


@pytest.mark.parametrize("shape", [(24, 24, 24), (9, 17, 70), (1, 5, 3), (33, 8, 129), (8, 9, 64), (17, 1, 1), (19, 37, 132)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["aniso", "iso", "het"])
def test_fused_volume_sweep_equals_three_kernels_and_oracle(shape, dtype, kind):
    """dim = 3: the one-pass sweep (k_fused3d: bricks of 8 x 8 x 64 with their forward neighbours in LDS) against the
    three-kernel path and the oracle, on shapes that do not fill whole bricks (examples/regularization.py:8-47 shape
    class; split_bregman_tvd.py:84-89,129-146, jacobi.py:41-54)."""
    rng = np.random.default_rng(21)
    vol = rng.random(shape).astype(dtype)
    mu = (0.03 + 0.2 * rng.random(shape)).astype(dtype) if kind == "het" else 0.1
    iso = kind == "iso"
    n = 5
    outs = {}
    for name, flags in (("fused3d", 0), ("three_kernel", _lib.FLAG_NO_FUSED)):
        with da.DeviceTVD(vol.shape, vol.dtype, dim=3) as st:
            st.set_coeffs(mu, 0.7, None, vol)
            st.upload(vol)
            res = st.run(n, isotropic=iso, flags=flags)
            outs[name] = st.download(want_split=True)
            launches = res.launches
        assert launches == (n if name == "fused3d" else 3 * n)
    for a, b in zip(outs["fused3d"], outs["three_kernel"]):
        assert_bits(a, b)
    ref = O.split_bregman_tvd(vol, mu=mu, omega=0.7, dim=3, max_num_iter=n, isotropic=iso, promotion="strict")
    assert_bits(outs["fused3d"][0], ref)


@pytest.mark.parametrize("channels", [2, 4])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("fuse_k", [1, 2, 3, 4])
def test_two_and_four_channel_images_run_the_temporally_blocked_sweep(monkeypatch, channels, dtype, fuse_k):
    """(H, W, 2) and (H, W, 4) arrays (all axes coupled as in jacobi.py:41) through the K-iterations-per-launch
    pipelines: same bits as the oracle, for iteration counts that are and are not multiples of K, with the eps
    test, anisotropic and isotropic; and K iterations really are one launch."""
    monkeypatch.setenv("DXH_FUSE_K", str(fuse_k))
    rng = np.random.default_rng(100 + channels)
    for shape in ((37, 300, channels), (130, 1030, channels)):
        img = rng.random(shape).astype(dtype)
        for n_iter, iso in ((fuse_k * 2, False), (fuse_k * 2 + 1, False), (5, True)):
            out, res = run_device(img, mu=0.1, n_iter=n_iter, isotropic=iso)
            assert_bits(out, O.split_bregman_tvd(img, mu=0.1, max_num_iter=n_iter, isotropic=iso, promotion="strict"))
            assert res.launches == -(-n_iter // fuse_k)
    img = rng.random((64, 300, channels)).astype(dtype)
    ref, _, _, done = O.split_bregman_tvd(img, mu=0.1, max_num_iter=60, eps=2e-3, promotion="strict", return_state=True)
    out, res = run_device(img, mu=0.1, n_iter=60, eps=2e-3)
    assert res.converged == 1 and res.iters_done == done and 1 < done < 60
    assert_bits(out, ref)


def test_square_root_chain_is_the_correctly_rounded_square_root_for_every_float():
    """The float kernels' square root (isotropic shrinkage, Chambolle's norm) is a five-instruction rsq chain on
    [2^-100, inf) and the IEEE sequence elsewhere; the device compares it with sqrtf on all 2^32 bit patterns."""
    import ctypes as C

    bad, chain = C.c_ulonglong(1), C.c_ulonglong(0)
    _lib.check(_lib.load().dxh_sqrt_probe(_lib.default_device(), C.byref(bad), C.byref(chain)))
    assert bad.value == 0
    assert chain.value == (0x7F800000 - 0x0D800000) + 1   # every float in [2^-100, inf) and +0
    # the shared reciprocal of Chambolle's three quotients: v_rcp_f32 + one Newton step == RN(1 / d) on [2^-100, 2^100]
    bad, n = C.c_ulonglong(1), C.c_ulonglong(0)
    _lib.check(_lib.load().dxh_rcp_probe(_lib.default_device(), C.byref(bad), C.byref(n)))
    assert bad.value == 0 and n.value == 0x71800000 - 0x0D800000 + 1


def test_isotropic_shrink_factor_reciprocal_route_is_exact():
    """The float kernels form max(s - k, 0) / (s + 1e-18) through a reciprocal seeded by the square root's own rsq
    (dxh_common.h: iso_s_fac_fast): that reciprocal is RN(1 / s) for every float s^2 in [2^-60, 2^60) (exhaustive), and
    the routine with its screens equals the IEEE expressions on 8.6e9 pseudo-random operand pairs."""
    import ctypes as C

    bad_r, bad_f, n = C.c_ulonglong(1), C.c_ulonglong(1), C.c_ulonglong(0)
    _lib.check(_lib.load().dxh_isofac_probe(_lib.default_device(), C.byref(bad_r), C.byref(bad_f), C.byref(n)))
    assert bad_r.value == 0 and bad_f.value == 0
    assert n.value == (0x5D800000 - 0x21800000) + 8192 * 256 * 4096


@pytest.mark.parametrize("shape", [(24, 24, 24), (9, 17, 70), (40, 33, 129), (3, 5, 2), (70, 26, 57), (1, 1, 9), (6, 9, 64), (13, 7, 132)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("maxiter", [2, 3, 4, 5, 7, 20])
def test_multi_sweep_jacobi_on_volumes(shape, dtype, maxiter):
    """Jacobi(maxiter > 1) with dim = 3 (examples/regularization.py:8-47 runs Jacobi(maxiter=20) on a volume) against the
    oracle's sweeps (jacobi.py:107-114), stand-alone and as the inner solver of the TV loop, on ragged shapes."""
    rng = np.random.default_rng(31)
    x0 = rng.random(shape).astype(dtype)
    rhs = rng.random(shape).astype(dtype)
    sol = da.Jacobi(maxiter=maxiter, dim=3, mass_coeff=0.7, diffusion_coeff=0.3)
    got = sol(x0, rhs)
    want = O.jacobi(x0, rhs, 0.7, 0.3, dim=3, maxiter=maxiter, promotion="strict")
    assert_bits(got, want)
    if maxiter in (3, 20) and shape != (1, 1, 9):
        out, res = run_device(x0, mu=0.1, omega=0.9, dim=3, n_iter=3, jm=maxiter)
        assert_bits(out, O.split_bregman_tvd(x0, mu=0.1, omega=0.9, dim=3, max_num_iter=3, jacobi_maxiter=maxiter,
                                             promotion="strict"))
        # rhs + sweeps + shrink per outer iteration; volumes of at least 16 x 32 in-plane cells sweep three (two) at a time
        # (dxh_vol_multi.h): 3 -> one pass, 20 -> 3 x 6 + 2 -> seven passes
        tiled = shape[1] >= 16 and shape[2] >= 32
        assert res.launches == 3 * (2 + ({3: 1, 20: 7}[maxiter] if tiled else maxiter))


def test_isotropic_relaxations_verified_per_threshold_change_no_bit(monkeypatch):
    """The K-stage isotropic float pipelines drop the per-element significand screen and stop the quotient after one residual
    correction when an exhaustive device check of every float x in [2^-60, 2^60) says so for the call's threshold
    (iso_split_vec / k_isolean_check): same bits as the screened five-operation form (DXH_FUSED_TUNE bit 16777216) and as the
    oracle, for thresholds of several magnitudes, incl. values that take the IEEE branch (zeros, huge, tiny)."""
    rng = np.random.default_rng(77)
    img = rng.random((96, 300, 3)).astype(np.float32)
    img[10:20, 30:60] = 0.0
    img[40, 100:120] *= 1e12     # squares beyond 2^60: the IEEE branch (no overflow: the results stay finite and comparable)
    img[50, 10:40] *= 1e-25      # squares below 2^-60: the IEEE branch
    for mu, ell in ((0.1, None), (0.37, 0.11), (0.05, 2.0), (1.0, 0.7)):
        lean, _ = run_device(img, mu=mu, ell=ell, n_iter=8, isotropic=True)
        monkeypatch.setenv("DXH_FUSED_TUNE", "16777216")
        screened, _ = run_device(img, mu=mu, ell=ell, n_iter=8, isotropic=True)
        monkeypatch.delenv("DXH_FUSED_TUNE")
        assert np.isfinite(lean).all()
        assert_bits(lean, screened)
        assert_bits(lean, O.split_bregman_tvd(img, mu=mu, ell=ell, max_num_iter=8, isotropic=True, promotion="strict"))


# ---- BASELINE.json full size (4000 x 8000 x 3 float32) -------------------------------------------------
def test_full_size_config2_parity_and_invariants(monkeypatch):
    """configs[1] at its real size: (a) 3 iterations bit-exact against the oracle over the whole image (the NumPy
    oracle needs ~3 s per iteration here); (b) 100 iterations: the K = 4 temporally blocked path and the
    one-iteration-per-launch path give identical bits; (c) the result is finite and has less total variation than
    the input; (d) the 100-iteration result equals the oracle's on four 512 x 512 windows (corners and centre)."""
    from darsia_amd.synthetic import synthetic_image

    img = synthetic_image(4000, 8000, 3)
    out3, _ = run_device(img, mu=0.1, n_iter=3)
    ref3 = O.split_bregman_tvd(img, mu=0.1, max_num_iter=3, promotion="strict")
    assert_bits(out3, ref3)
    ref3_np2 = O.split_bregman_tvd(img, mu=0.1, max_num_iter=3)  # reference dtype flow
    np.testing.assert_allclose(out3, ref3_np2, rtol=RTOL, atol=ATOL)
    out_k4, res = run_device(img, mu=0.1, n_iter=100)
    monkeypatch.setenv("DXH_FUSE_K", "1")
    out_k1, _ = run_device(img, mu=0.1, n_iter=100)
    assert res.iters_done == 100
    assert_bits(out_k4, out_k1)
    # the public host-array call takes the row-band path at this size (dxh_tvd_solve_banded): same bits, and so the
    # window checks (d) below pin what a caller of split_bregman_tvd() gets
    monkeypatch.delenv("DXH_FUSE_K")
    assert_bits(da.split_bregman_tvd(img, mu=0.1, max_num_iter=100), out_k4)
    # (the reference's channel-coupled Jacobi sum is not a convex combination -- 6 neighbours over a diagonal
    # built for 4, quirk Q2 -- so the colour result is brighter than the input: no range check here)
    assert np.isfinite(out_k4).all()
    # TV denoising reduces the total variation of the image
    tv = lambda a: np.abs(np.diff(a, axis=0)).sum(dtype=np.float64) + np.abs(np.diff(a, axis=1)).sum(dtype=np.float64)  # noqa: E731
    assert tv(out_k4) < tv(img)
    # (d) the 100-iteration result itself against the oracle, through windows.  u_{n+1}(x) depends on u_n within one
    # pixel of x (and on d_n, b_n at x - 1 .. x, which depend on u_n up to x + 1; split_bregman_tvd.py:84-89, 129-146,
    # jacobi.py:41-54), so after n iterations a pixel knows nothing beyond n + 1 pixels: the oracle run on a 512 x 512
    # crop equals the full-size result wherever the crop's cut edges are more than 101 pixels away (a crop edge that IS
    # the image edge carries the same boundary rule and needs no margin).  Strict flow: bit for bit; the reference's
    # NumPy-2 float64 promotion: the north-star tolerance rtol 1e-4.
    H, W, n, m = 4000, 8000, 100, 102
    for (r0, c0) in ((0, 0), ((H - 512) // 2, (W - 512) // 2), (H - 512, W - 512), (0, W - 512)):
        crop = np.ascontiguousarray(img[r0:r0 + 512, c0:c0 + 512])
        ra = 0 if r0 == 0 else m
        rb = 512 if r0 + 512 == H else 512 - m
        ca = 0 if c0 == 0 else m
        cb = 512 if c0 + 512 == W else 512 - m
        want = O.split_bregman_tvd(crop, mu=0.1, max_num_iter=n, promotion="strict")[ra:rb, ca:cb]
        got = out_k4[r0 + ra:r0 + rb, c0 + ca:c0 + cb]
        assert_bits(got, want)
        want2 = O.split_bregman_tvd(crop, mu=0.1, max_num_iter=n)[ra:rb, ca:cb]  # reference dtype flow
        np.testing.assert_allclose(got, want2, rtol=RTOL, atol=ATOL)


def test_verbose_prints_increment_and_energy(golden, capsys):
    """verbose=True prints the reference's two messages; the energy is the functional of
    split_bregman_tvd.py:71-77 (compared here with a NumPy evaluation)."""
    img = golden["g2_aniso2d_f32/img"]
    out = da.split_bregman_tvd(img, mu=0.1, max_num_iter=3, verbose=True)
    text = capsys.readouterr().out.splitlines()
    assert text[0].startswith("The energy functional starts at ")
    assert len([ln for ln in text if ln.startswith("Split Bregman iteration")]) == 3
    assert_bits(out, O.split_bregman_tvd(img, mu=0.1, max_num_iter=3, promotion="strict"))

    def functional(x):
        x = x.astype(np.float64)
        return 0.5 * np.sum((x - img) ** 2) + sum(np.abs(0.1 * O.backward_diff(x, j)).sum() for j in range(2))

    e0 = float(text[0].split()[-1])
    assert abs(e0 - functional(img)) <= 1e-6 * functional(img)
    u2 = O.split_bregman_tvd(img, mu=0.1, max_num_iter=2, promotion="strict")
    e_last = float(text[-1].split()[-1])  # energy of the iterate BEFORE the third update
    assert abs(e_last - functional(u2)) <= 1e-5 * functional(u2)


@pytest.mark.parametrize("name,kw,sched", [
    ("g14_adaptive_f64", dict(mu=0.1, max_num_iter=9), lambda it: it % 3 == 1),
    ("g14_adaptive_f32", dict(mu=0.1, max_num_iter=9), lambda it: it % 3 == 1),
    ("g14_adaptive_rgb_f32", dict(mu=0.1, omega=0.9, max_num_iter=6), lambda it: it >= 2),
])
def test_adaptive_regularisation(golden, name, kw, sched):
    """adaptive ell (split_bregman_tvd.py:149-158) incl. the reference's quirk that the Jacobi diagonal keeps
    its first value: explicit (d, b) planes on the device, schedule evaluated on the host."""
    img = golden[name + "/img"]
    out = da.split_bregman_tvd(img, adaptive=sched, **kw)
    assert_bits(out, O.split_bregman_tvd(img, adaptive=sched, promotion="strict", **kw))
    if img.dtype == np.float64:
        assert_bits(out, golden[name + "/out"])
    else:
        # ell = 1/|grad u| makes the iteration expansive (values grow to 1e7 on a [0, 1] image after 9
        # iterations), so the reference's float64 promotion of u and a pure float32 flow decorrelate in a few
        # elements; bit-exactness against the same-dtype oracle above is the parity statement, here only a
        # robust closeness check against the reference's mixed flow
        ref = golden[name + "/out"].astype(np.float64)
        rel = np.abs(out - ref) / (np.abs(ref) + 1.0)
        assert np.median(rel) < 1e-4 and np.mean(rel < 1e-2) > 0.98


def test_adaptive_regularisation_with_an_array_valued_mu():
    """adaptive ell on top of an array-valued mu with scalar omega / ell: until the first update ell is a scalar next to
    the threshold plane (the kernels then read it as one), afterwards it varies per element -- 2-D and a volume, against
    the oracle (split_bregman_tvd.py:149-158)."""
    rng = np.random.default_rng(77)
    for shape, dim in (((40, 68), 2), ((9, 12, 64), 3)):
        img = rng.random(shape)
        mu = 0.05 + 0.2 * rng.random(shape)
        sched = lambda it: it in (1, 3)   # noqa: E731
        kw = dict(mu=mu, omega=0.8, ell=0.6, dim=dim, max_num_iter=6)
        assert_bits(da.split_bregman_tvd(img, adaptive=sched, **kw), O.split_bregman_tvd(img, adaptive=sched, promotion="strict", **kw))


@pytest.mark.parametrize("iso", [False, True])
@pytest.mark.parametrize("n_iter", [3, 7, 8])
def test_heterogeneous_temporal_blocking_bit_exact(iso, n_iter, monkeypatch):
    """Array-valued mu / omega: the K-iteration pipeline derives the coefficients in the kernel from the given planes;
    same bits as single-iteration sweeps reading the four derived planes, and as the oracle."""
    from darsia_amd.synthetic import synthetic_image
    img = synthetic_image(203, 517, 3)
    rng = np.random.default_rng(11)
    mu = (0.05 + 0.3 * rng.random(img.shape)).astype(np.float32)
    omega = (0.5 + rng.random(img.shape)).astype(np.float32)
    kw = dict(mu=mu, omega=omega, max_num_iter=n_iter, isotropic=iso)
    out = da.split_bregman_tvd(img, **kw)
    monkeypatch.setenv("DXH_FUSE_K", "1")
    monkeypatch.setenv("DXH_FUSED_TUNE", "32")
    ref = da.split_bregman_tvd(img, **kw)
    np.testing.assert_array_equal(out, ref)
    if not iso:
        assert_bits(out, O.split_bregman_tvd(img, promotion="strict", **kw))


@pytest.mark.parametrize("eps,expect", [(5e-3, 4), (2e-3, 6), (1e-3, 8), (5e-4, 13)])
def test_heterogeneous_eps_stop_with_temporal_blocking(eps, expect):
    """The stop iteration falls at every position inside the 3-iteration launches."""
    rng = np.random.default_rng(5)
    y, x = np.mgrid[0:160, 0:240]
    img = (0.5 + 0.25 * np.sign(np.sin(x / 20.0)) * np.sign(np.sin(y / 15.0)) + rng.normal(0, 0.01, (160, 240))).astype(np.float32)
    mu = np.full(img.shape, 0.02, np.float32)
    mu[:, 120:] = 0.05
    out = da.split_bregman_tvd(img, mu=mu, max_num_iter=60, eps=eps)
    ref, _, _, done = O.split_bregman_tvd(img, mu=mu, max_num_iter=60, eps=eps, promotion="strict", return_state=True)
    assert done == expect
    assert_bits(out, ref)


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (7, 1), (2, 2), (3, 250), (250, 3), (1, 1, 3), (2, 5, 3), (5, 1, 3), (9, 2, 1)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_degenerate_shapes_all_paths(shape, dtype):
    """One-pixel, one-row, one-column and two-wide images: every path (4-iteration pipeline, heterogeneous pipeline,
    eps, isotropic, Chambolle, Gauss-Seidel Bregman) against its oracle, bit for bit."""
    from oracle import bregman_gs_numpy as BG
    from oracle import chambolle_numpy as CH
    rng = np.random.default_rng(sum(shape))
    img = rng.random(shape).astype(dtype)
    for kw in (dict(mu=0.1, max_num_iter=9), dict(mu=0.05, max_num_iter=6, isotropic=True), dict(mu=0.07, max_num_iter=20, eps=1e-2)):
        iso = kw.get("isotropic", False)
        ref = O.split_bregman_tvd(img, promotion="strict", **kw)
        out = da.split_bregman_tvd(img, **kw)
        if iso:
            np.testing.assert_allclose(out, ref, rtol=1e-6 if dtype == np.float32 else 1e-13, atol=1e-7 if dtype == np.float32 else 1e-15)
        else:
            assert_bits(out, ref)
    mu = (0.03 + 0.1 * rng.random(shape)).astype(dtype)
    assert_bits(da.split_bregman_tvd(img, mu=mu, max_num_iter=7), O.split_bregman_tvd(img, mu=mu, max_num_iter=7, promotion="strict"))
    if min(shape[:2]) >= 1 and len(shape) <= 3:
        np.testing.assert_array_equal(da.denoise_tv_chambolle(img, weight=0.1, eps=0.0, max_num_iter=6),
                                      CH.denoise_tv_chambolle(img, weight=0.1, eps=0.0, max_num_iter=6))
        np.testing.assert_array_equal(da.denoise_tv_bregman(img, weight=1.0, max_num_iter=5, eps=0.0, isotropic=False),
                                      BG.denoise_tv_bregman(img, weight=1.0, max_num_iter=5, eps=0.0, isotropic=False))


def test_reference_regularization_example_flow(capsys):
    """examples/regularization.py on a synthetic volume: heterogeneous 3-D TVD with Jacobi(maxiter=20), verbose, eps,
    through ``tvd(img=Image, ...)``; then H1_regularization with a tolerance-controlled Jacobi."""
    rng = np.random.default_rng(8)
    z, y, x = np.mgrid[0:36, 0:44, 0:30]
    vol = (0.5 + 0.3 * np.sign(np.sin(x / 6.0)) * np.sign(np.sin(y / 7.0)) * np.sign(np.sin(z / 5.0)) + rng.normal(0, 0.05, (36, 44, 30)))
    img = da.Image(vol, space_dim=3, dimensions=[0.1, 0.1, 0.1])
    mu = np.ones_like(img.img)
    mu[:, 0:20, :] = 0.5
    out = da.tvd(img=img, method="heterogeneous bregman", isotropic=False, weight=mu, omega=0.1, max_num_iter=30, eps=1e-6,
                 dim=3, verbose=True, solver=da.Jacobi(maxiter=20))
    assert isinstance(out, da.Image) and out.img.shape == vol.shape and out.img is not img.img
    assert "relative increment" in capsys.readouterr().out
    # TVD forwards ell = regularization, whose default is 1.0 (tvd.py:37,111), not 2 mu
    ref = O.split_bregman_tvd(vol, mu=mu, omega=0.1, ell=1.0, max_num_iter=30, eps=1e-6, dim=3, jacobi_maxiter=20,
                              promotion="strict")
    assert_bits(out.img, ref)
    h1 = da.H1_regularization(img=img, mu=1, omega=1, dim=3, solver=da.Jacobi(maxiter=30, tol=1e-6))
    ref_h1 = O.h1_regularization(vol, mu=1, omega=1, dim=3, jacobi_maxiter=30, jacobi_tol=1e-6, promotion="strict")
    assert_bits(h1.img, ref_h1)


@pytest.mark.parametrize("shape", [(90, 130), (60, 85, 3)])
def test_chambolle_stop_iteration_inside_two_iteration_passes(shape):
    """Every stop iteration from 2 to 12 (both positions inside the two-iteration passes, and the single-iteration
    tail when max_num_iter ends the run): same iteration count and bits as the oracle."""
    from oracle import chambolle_numpy as CH
    from darsia_amd.synthetic import synthetic_image
    img = synthetic_image(shape[0], shape[1], 3).astype(np.float64)
    img = np.ascontiguousarray(img if len(shape) == 3 else img[..., 0])
    seen = set()
    for eps in (3e-2, 2e-2, 1e-2, 5e-3, 2e-3, 1e-3, 7e-4, 5e-4, 3e-4, 2e-4):
        ref, n_ref = CH.denoise_tv_chambolle(img, weight=0.12, eps=eps, max_num_iter=50, return_iters=True)
        out, n = da.denoise_tv_chambolle(img, weight=0.12, eps=eps, max_num_iter=50, return_iters=True)
        assert n == n_ref
        np.testing.assert_array_equal(out, ref)
        seen.add(n)
    # the gray image stops at odd counts (second iteration of a pass), the colour image also at 4 (first iteration)
    assert ({3, 5, 7, 11} <= seen) if len(shape) == 2 else ({3, 4, 5} <= seen)
    for m in (1, 2, 3, 4, 5, 6, 7):  # runs ended by max_num_iter
        ref, n_ref = CH.denoise_tv_chambolle(img, weight=0.12, eps=0.0, max_num_iter=m, return_iters=True)
        out, n = da.denoise_tv_chambolle(img, weight=0.12, eps=0.0, max_num_iter=m, return_iters=True)
        assert n == n_ref == m
        np.testing.assert_array_equal(out, ref)


@pytest.mark.parametrize("maxiter", [2, 3, 4, 5, 6, 7, 9, 20])
@pytest.mark.parametrize("shape,dtype", [((70, 300, 3), np.float32), ((61, 333), np.float64), ((1, 40), np.float32), ((9, 2, 3), np.float64)])
def test_multi_sweep_jacobi_bit_exact(maxiter, shape, dtype, monkeypatch):
    """Jacobi(maxiter > 1): several sweeps per pass (k_jacobi_multi) against the sweep-by-sweep kernel and the oracle,
    stand-alone, inside the TV loop and through H1_regularization."""
    rng = np.random.default_rng(maxiter)
    x0 = rng.random(shape).astype(dtype)
    rhs = rng.random(shape).astype(dtype)
    fast = da.Jacobi(maxiter=maxiter, mass_coeff=1.0, diffusion_coeff=0.4, dim=2)(x0, rhs)
    tv_fast = da.split_bregman_tvd(x0, mu=0.1, max_num_iter=3, solver=da.Jacobi(maxiter=maxiter))
    h1_fast = da.H1_regularization(x0, mu=0.3, omega=1.0, dim=2, solver=da.Jacobi(maxiter=maxiter))
    monkeypatch.setenv("DXH_FUSED_TUNE", "64")  # sweep by sweep
    np.testing.assert_array_equal(fast, da.Jacobi(maxiter=maxiter, mass_coeff=1.0, diffusion_coeff=0.4, dim=2)(x0, rhs))
    assert_bits(fast, O.jacobi(x0, rhs, 1.0, 0.4, dim=2, maxiter=maxiter, promotion="strict"))
    assert_bits(tv_fast, O.split_bregman_tvd(x0, mu=0.1, max_num_iter=3, jacobi_maxiter=maxiter, promotion="strict"))
    assert_bits(h1_fast, O.h1_regularization(x0, mu=0.3, omega=1.0, dim=2, jacobi_maxiter=maxiter, promotion="strict"))


def test_regressions_found_by_the_fuzzer():
    """(1) (H, W, 1) arrays sum over their length-1 third axis in every sweep of Jacobi(maxiter > 1);
    (2) array-valued mu with scalar omega and ell: the diagonal is a Python float folded in double."""
    rng = np.random.default_rng(21)
    img = rng.random((18, 27, 1))
    assert_bits(da.split_bregman_tvd(img, mu=0.1, ell=0.3, max_num_iter=3, solver=da.Jacobi(maxiter=2)),
                O.split_bregman_tvd(img, mu=0.1, ell=0.3, max_num_iter=3, jacobi_maxiter=2, promotion="strict"))
    for shape in ((68, 331, 3), (61, 159), (29, 24, 3), (13, 617, 2)):
        img = rng.random(shape).astype(np.float32)
        mu = (0.02 + 0.2 * rng.random(shape)).astype(np.float32)
        for n in (1, 4, 7):
            assert_bits(da.split_bregman_tvd(img, mu=mu, ell=0.3, max_num_iter=n),
                        O.split_bregman_tvd(img, mu=mu, ell=0.3, max_num_iter=n, promotion="strict"))
    # the same (only mu varies: constant Jacobi divisors inside the array-coefficient pipeline) in float64,
    # isotropic, and with the stopping test
    img = rng.random((45, 333, 3))
    mu = 0.02 + 0.2 * rng.random(img.shape)
    for iso in (False, True):
        for n in (3, 8):
            assert_bits(da.split_bregman_tvd(img, mu=mu, omega=1.3, ell=0.7, max_num_iter=n, isotropic=iso),
                        O.split_bregman_tvd(img, mu=mu, omega=1.3, ell=0.7, max_num_iter=n, isotropic=iso, promotion="strict"))
    img32, mu32 = img.astype(np.float32), mu.astype(np.float32)
    assert_bits(da.split_bregman_tvd(img32, mu=mu32, ell=0.7, max_num_iter=30, eps=1e-3),
                O.split_bregman_tvd(img32, mu=mu32, ell=0.7, max_num_iter=30, eps=1e-3, promotion="strict"))


def test_negative_threshold_and_exact_zeros():
    """The fused kernels shrink through q - clamp(q, -k, k); a negative k = mu / ell (not meaningful, but the
    reference evaluates it) must take the generic path, and images with exact zeros / flat regions (q == 0,
    |q| == k) must agree bit for bit."""
    rng = np.random.default_rng(5)
    for dt in (np.float32, np.float64):
        img = rng.random((40, 300, 3)).astype(dt)
        img[10:20] = 0.5
        img[:, 100:140] = 0.0
        for mu in (-0.1, 0.0, 0.25):
            for iso in (False, True):
                for n in (1, 4, 9):
                    assert_bits(da.split_bregman_tvd(img, mu=mu, ell=0.5, max_num_iter=n, isotropic=iso),
                                O.split_bregman_tvd(img, mu=mu, ell=0.5, max_num_iter=n, isotropic=iso, promotion="strict"))
    # steps of exactly 2k: after the first solve several q sit exactly on the threshold
    img = np.zeros((32, 256), np.float32)
    img[:, 128:] = 0.5
    for n in (1, 2, 5, 8):
        assert_bits(da.split_bregman_tvd(img, mu=0.25, ell=1.0, max_num_iter=n),
                    O.split_bregman_tvd(img, mu=0.25, ell=1.0, max_num_iter=n, promotion="strict"))


def test_small_random_sweep_of_parameters():
    """A short run of tests/fuzz/fuzz_tvd.py's generator inside the suite."""
    rng = np.random.default_rng(1234)
    for _ in range(60):
        h, w = int(rng.integers(1, 60)), int(rng.integers(1, 500))
        c = int(rng.choice([0, 1, 2, 3, 4]))
        shape = (h, w) if c == 0 else (h, w, c)
        dt = rng.choice([np.float32, np.float64])
        img = rng.random(shape).astype(dt)
        kw = dict(max_num_iter=int(rng.integers(1, 12)))
        het = rng.random() < 0.4
        kw["mu"] = (0.02 + 0.2 * rng.random(shape)).astype(dt) if het else float(rng.choice([0.05, 0.1, 0.3]))
        if rng.random() < 0.3:
            kw["omega"] = (0.5 + rng.random(shape)).astype(dt) if (het and rng.random() < 0.5) else float(rng.choice([0.5, 2.0]))
        if rng.random() < 0.3:
            kw["ell"] = float(rng.choice([0.3, 1.0]))
        if rng.random() < 0.3:
            kw["eps"] = float(rng.choice([1e-2, 3e-3]))
        jm = int(rng.choice([1, 1, 2, 3, 5]))
        assert_bits(da.split_bregman_tvd(img, solver=da.Jacobi(maxiter=jm), **kw),
                    O.split_bregman_tvd(img, jacobi_maxiter=jm, promotion="strict", **kw))


def test_chambolle_single_row_wider_than_a_block():
    """A one-row image with more marching waves than element-wise blocks (partial-sum buffer sizing; found by the fuzzer)."""
    from oracle import chambolle_numpy as CH
    rng = np.random.default_rng(4)
    for shape in ((1, 250), (1, 300, 3), (2, 900)):
        img = rng.random(shape)
        np.testing.assert_array_equal(da.denoise_tv_chambolle(img, weight=0.1, eps=0.0, max_num_iter=5),
                                      CH.denoise_tv_chambolle(img, weight=0.1, eps=0.0, max_num_iter=5))


def test_c_abi_in_place_output_is_safe():
    """dxh_chambolle with out == image (in-place use of the C ABI) on an array big enough for the page-touch thread:
    the input must not be clobbered before it is uploaded."""
    import ctypes as C
    from darsia_amd import _lib
    from darsia_amd.synthetic import synthetic_image
    img = synthetic_image(1200, 2000, 3)  # 28.8 MB > the 8 MB threshold
    ref = da.denoise_tv_chambolle(img, weight=0.1, eps=0.0, max_num_iter=3)
    buf = img.copy()
    done = C.c_int(0)
    _lib.check(_lib.load().dxh_chambolle(_lib.default_device(), 0, 3, 1200, 2000, 3, _lib.ptr(buf), _lib.ptr(buf), 0.1, 0.0, 3,
                                          C.byref(done)))
    np.testing.assert_array_equal(buf, ref)


def test_division_check_detects_wrong_quotients_and_both_paths_agree(monkeypatch):
    """The float pipelines stop the constant division after ONE residual correction when an exhaustive device
    test says that is exact for the coefficient set.  (1) The test itself: exact reciprocals of ordinary
    divisors pass, a visibly wrong reciprocal must be caught.  (2) DXH_FUSED_TUNE=128 disables the
    shortcut; both paths give the oracle's bits."""
    import ctypes as C
    from darsia_amd import _lib
    lib = _lib.load()
    bad = C.c_uint(7)
    _lib.check(lib.dxh_div3_probe(0, 1.8, 0.0, 9.0, 0.0, C.byref(bad)))
    assert bad.value == 0
    # one correction squares the reciprocal's relative error: 2^-17 off leaves 2^-34, which flips the rounding of
    # thousands of the 2^23 quotients
    r_off = float(np.float32(1) / np.float32(1.8) * np.float32(1 + 2.0 ** -17))
    _lib.check(lib.dxh_div3_probe(0, 1.8, r_off, 9.0, 0.0, C.byref(bad)))
    assert bad.value == 1
    # (a small integer divisor like 9 would not do: its quotients stay 1/18 ulp away from every rounding boundary)
    r_off = float(np.float32(1) / np.float32(1.3) * np.float32(1 - 2.0 ** -17))
    _lib.check(lib.dxh_div3_probe(0, 1.8, 0.0, 1.3, r_off, C.byref(bad)))
    assert bad.value == 2
    rng = np.random.default_rng(11)
    img = rng.random((50, 420, 3)).astype(np.float32)
    want = O.split_bregman_tvd(img, mu=0.1, max_num_iter=9, promotion="strict")
    assert_bits(da.split_bregman_tvd(img, mu=0.1, max_num_iter=9), want)
    monkeypatch.setenv("DXH_FUSED_TUNE", "128")
    assert_bits(da.split_bregman_tvd(img, mu=0.1, max_num_iter=9), want)


def test_planes_larger_than_four_gigabytes():
    """Maximum sizes: a 36000 x 32000 float32 image has 4.6 GB planes, i.e. byte offsets beyond 2^32 and element offsets
    beyond 2^30 inside every kernel of the sweep.  Four iterations (one K = 4 launch) and the generic three-kernel path,
    pinned to the oracle on windows at the far corner, the near corner and across the 4 GB line (dependence radius n + 1)."""
    H, W, n = 36000, 32000, 4
    rng = np.random.default_rng(77)
    img = rng.random((H, W), dtype=np.float32)
    m = n + 2
    r4g = (1 << 32) // (4 * W)  # the row in which the 2^32-th byte of a plane lies
    windows = [(0, 0), (H - 256, W - 256), (r4g - 128, W // 2)]
    refs = []
    for (r0, c0) in windows:
        crop = np.ascontiguousarray(img[r0:r0 + 256, c0:c0 + 256])
        refs.append(O.split_bregman_tvd(crop, mu=0.1, max_num_iter=n, promotion="strict"))
    for flags in (0, _lib.FLAG_NO_FUSED):
        out, res = run_device(img, mu=0.1, n_iter=n, flags=flags)
        assert res.launches == (1 if flags == 0 else 3 * n)
        for (r0, c0), ref in zip(windows, refs):
            ra, rb = (0 if r0 == 0 else m), (256 if r0 + 256 == H else 256 - m)
            ca, cb = (0 if c0 == 0 else m), (256 if c0 + 256 == W else 256 - m)
            assert_bits(out[r0 + ra:r0 + rb, c0 + ca:c0 + cb], ref[ra:rb, ca:cb])
        del out


# ---- whole call in row bands (dxh_tvd_solve_banded) ------------------------------------------------------------
def _one_piece(img, mu, omega, ell, n_iter, m=1, iso=False):
    with da.DeviceTVD(img.shape, img.dtype, dim=2) as st:
        st.set_coeffs(mu, omega, ell, img)
        st.upload(img)
        st.run(n_iter, m, None, eps=None, isotropic=iso)
        return st.download()


@pytest.mark.parametrize("shape,nbands,n_iter,m", [
    ((200, 96), 2, 7, 1), ((200, 96, 3), 3, 9, 1), ((333, 70, 3), 5, 4, 1), ((257, 130), 4, 3, 2),
    ((64, 40, 3), 8, 30, 1),      # halos longer than the bands: every band holds nearly the whole image
    ((97, 33, 2), 3, 5, 1), ((120, 50, 3), 1, 6, 1),
])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_banded_whole_call_is_bit_identical(shape, nbands, n_iter, m, dtype):
    from darsia_amd.restoration.split_bregman_tvd import solve_banded

    rng = np.random.default_rng(hash((shape, nbands)) % 2**32)
    img = rng.random(shape).astype(dtype)
    want = _one_piece(img, 0.1, 1.0, None, n_iter, m)
    got, used = solve_banded(img, 0.1, 1.0, None, n_iter, m, nbands=nbands, return_bands=True)
    assert used == min(nbands, max(1, shape[0] // 8)) and got.dtype == img.dtype
    assert np.array_equal(got, want)


def test_banded_whole_call_variants_and_errors():
    from darsia_amd.restoration.split_bregman_tvd import solve_banded

    rng = np.random.default_rng(77)
    img = rng.random((240, 88, 3)).astype(np.float32)
    # isotropic, array-valued mu / omega / ell (the bands take their rows of the arrays)
    assert np.array_equal(solve_banded(img, 0.1, 1.0, None, 6, isotropic=True, nbands=3), _one_piece(img, 0.1, 1.0, None, 6, iso=True))
    mu = (0.05 + 0.1 * rng.random(img.shape)).astype(np.float32)
    om = (0.5 + rng.random(img.shape)).astype(np.float32)
    ell = (0.1 + 0.2 * rng.random(img.shape)).astype(np.float32)
    assert np.array_equal(solve_banded(img, mu, 1.0, None, 5, nbands=4), _one_piece(img, mu, 1.0, None, 5))
    assert np.array_equal(solve_banded(img, mu, om, ell, 5, nbands=2), _one_piece(img, mu, om, ell, 5))
    assert np.array_equal(solve_banded(img, 0.2, 2.0, 0.3, 5, nbands=3), _one_piece(img, 0.2, 2.0, 0.3, 5))
    # the automatic choice keeps a small image in one piece
    out, used = solve_banded(img, 0.1, 1.0, None, 5, return_bands=True)
    assert used == 1 and np.array_equal(out, _one_piece(img, 0.1, 1.0, None, 5))
    with pytest.raises(ValueError):
        solve_banded(img, 0.1, 1.0, None, 5, nbands=65)
    with pytest.raises(ValueError):
        solve_banded(img, 0.1, 1.0, 0.0, 5)   # ell == 0


def test_split_bregman_tvd_takes_the_banded_path_for_large_host_images():
    """32 MB and more, fixed iteration count: split_bregman_tvd() goes through dxh_tvd_solve_banded; same values as the
    device-resident run, and as the oracle on a window (dependence radius n + 1)."""
    from oracle import tvd_numpy as O

    rng = np.random.default_rng(5)
    img = rng.random((2048, 1600, 3)).astype(np.float32)   # 39 MB
    got = da.split_bregman_tvd(img, mu=0.1, max_num_iter=12)
    assert np.array_equal(got, _one_piece(img, 0.1, 1.0, None, 12))
    win = img[1000:1096, 700:780]
    ref = O.split_bregman_tvd(win, mu=0.1, max_num_iter=12, promotion="strict")
    assert np.array_equal(got[1000:1096, 700:780][14:-14, 14:-14], ref[14:-14, 14:-14])


def test_plane_cache_reuses_and_trims():
    """Released context planes are kept for exact-size re-use (dxh_common.h: PlanePool); dxh_mem_trim hands them back.
    A re-used plane comes back cleared: results do not depend on what ran before."""
    rng = np.random.default_rng(11)
    img = rng.random((300, 200, 3)).astype(np.float32)
    da.trim_device_cache()
    first = _one_piece(img, 0.1, 1.0, None, 5)
    other = _one_piece(rng.random((300, 200, 3)).astype(np.float32), 0.3, 2.0, 0.7, 9, iso=True)   # dirties the cached planes
    assert other.shape == img.shape
    again = _one_piece(img, 0.1, 1.0, None, 5)
    assert np.array_equal(first, again)
    released = da.trim_device_cache()
    assert released >= 7 * 300 * 608 * 4    # of, u x2, q x4 at least (pitch 608)
    assert da.trim_device_cache() == 0


def test_operator_built_from_a_workflow_configuration_runs_the_heterogeneous_pipeline():
    """presets/workflows/restoration.py:60-151: porosity weight + ignore mask -> TVD("heterogeneous bregman") with an
    array-valued weight; the operator's result equals the oracle's for the same coefficients (float64: bit for bit)."""
    from types import SimpleNamespace

    from darsia_amd.presets.workflows import build_restoration
    from darsia_amd.presets.workflows.config import RestorationConfig
    from oracle import tvd_numpy as O

    rng = np.random.default_rng(21)
    h, w = 40, 56
    porosity = 0.2 + 0.6 * rng.random((h, w))
    inside = np.ones((h, w), bool)
    inside[:, :9] = False
    rig = SimpleNamespace(baseline=da.ScalarImage(np.zeros((h, w))), image_porosity=da.ScalarImage(porosity),
                          boolean_porosity=da.ScalarImage(inside))
    cfg = RestorationConfig().load({"method": "tvd", "ignore": ["boolean_porosity"],
                                    "options": {"weight": "image_porosity", "max_num_iter": 9, "eps": 1e-12,
                                                "omega": 1.5, "regularization": 0.4}})
    op = build_restoration(cfg, rig)
    assert op.method == "heterogeneous bregman"
    img = rng.random((h, w))
    got = op(img)
    want = O.split_bregman_tvd(img, mu=porosity * inside, omega=1.5, ell=0.4, max_num_iter=9, eps=1e-12, promotion="strict")
    assert got.dtype == np.float64 and np.array_equal(got, want)


def test_coefficient_arrays_of_the_other_float_type_are_converted_on_the_device():
    """A float64 weight map with a float32 image (the workflow's porosity weights) and the reverse: same result as
    converting the map with ndarray.astype first; omega and ell arrays too; broadcastable shapes still work."""
    rng = np.random.default_rng(31)
    img = rng.random((70, 90, 3)).astype(np.float32)
    mu64 = 0.05 + 0.1 * rng.random(img.shape)
    om64 = 0.5 + rng.random(img.shape)
    ell64 = 0.1 + 0.3 * rng.random(img.shape)
    got = da.split_bregman_tvd(img, mu=mu64, omega=om64, ell=ell64, max_num_iter=6)
    want = da.split_bregman_tvd(img, mu=mu64.astype(np.float32), omega=om64.astype(np.float32), ell=ell64.astype(np.float32), max_num_iter=6)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    img64 = img.astype(np.float64)
    mu32 = mu64.astype(np.float32)
    assert np.array_equal(da.split_bregman_tvd(img64, mu=mu32, max_num_iter=5),
                          da.split_bregman_tvd(img64, mu=mu32.astype(np.float64), max_num_iter=5))
    gray = np.ascontiguousarray(img[..., 0])
    col = (0.05 + 0.1 * rng.random((1, 90)))                     # float64, broadcast along the rows
    assert np.array_equal(da.split_bregman_tvd(gray, mu=col, max_num_iter=5),
                          da.split_bregman_tvd(gray, mu=np.broadcast_to(col, gray.shape).astype(np.float32), max_num_iter=5))


@pytest.mark.parametrize("shape", [(150, 260), (141, 230, 3)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("iso", [False, True])
def test_array_mu_with_scalar_omega_ell_runs_the_per_element_threshold_pipeline(shape, dtype, iso, monkeypatch):
    """mu array, scalar omega and ell (the TVD operator with a weight map): k_fused_multi<..., KQV> -- the scalar pipeline
    with kq = mu / ell per element.  Bit for bit equal to the oracle, to the general array pipeline
    (DXH_FUSED_TUNE=65536) and, with a stopping criterion, stopping at the same iteration."""
    rng = np.random.default_rng(90)
    img = rng.random(shape).astype(dtype)
    mu = (0.02 + 0.25 * rng.random(shape)).astype(dtype)
    mu[5:20, 10:40] = 0.0                                    # zero weight: no shrinkage there
    kw = dict(mu=mu, omega=1.3, ell=0.35)
    for n in (1, 2, 3, 7, 12):
        out, _ = run_device(img, n_iter=n, isotropic=iso, **kw)
        if iso and img.ndim == 3:
            ref = O.split_bregman_tvd(img, max_num_iter=n, isotropic=True, promotion="strict", **kw)
        else:
            ref = O.split_bregman_tvd(img, max_num_iter=n, isotropic=iso, promotion="strict", **kw)
        assert_bits(out, ref)
    out_eps, res = run_device(img, n_iter=60, eps=3e-3, isotropic=iso, **kw)
    monkeypatch.setenv("DXH_FUSED_TUNE", "65536")
    general, res_g = run_device(img, n_iter=60, eps=3e-3, isotropic=iso, **kw)
    assert res.iters_done == res_g.iters_done and 1 < res.iters_done < 60
    assert_bits(out_eps, general)


@pytest.mark.parametrize("shape,dim", [((60, 103), 2), ((44, 103, 3), 2), ((17, 44, 103), 3)])
@pytest.mark.parametrize("mu,ell", [(0.1, 0.3), (0.7, 0.9), (0.05, None)])
def test_scalar_threshold_next_to_an_array_omega_is_rounded_once(shape, dim, mu, ell):
    """float32 image, float32 array omega, scalar mu and ell: the threshold mu / ell is a Python float in the reference
    (split_bregman_tvd.py:80-81,143), rounded to float32 ONCE where it meets the float32 array -- float32(0.1) / float32(0.3) is
    one ulp off float32(0.1 / 0.3).  Found by tests/fuzz/fuzz_r4.py (the derived-coefficient path divided the rounded scalars);
    both the native flow and the default call (which takes the native flow for float32 coefficient arrays) against the oracle."""
    rng = np.random.default_rng(5)
    img = rng.random(shape).astype(np.float32)
    om = (0.5 + rng.random(shape)).astype(np.float32)
    kw = dict(mu=mu, omega=om, dim=dim, max_num_iter=4)
    if ell is not None:
        kw["ell"] = ell
    ref = O.split_bregman_tvd(img, promotion="strict", **kw)
    assert np.array_equal(O.split_bregman_tvd(img, promotion="numpy2", **kw), ref)   # float32 arrays only: one flow
    assert_bits(da.split_bregman_tvd(img, promotion="native", **kw), ref)
    assert_bits(da.split_bregman_tvd(img, **kw), ref)
    assert_bits(da.split_bregman_tvd(img, solver=da.Jacobi(maxiter=3), **kw),
                O.split_bregman_tvd(img, jacobi_maxiter=3, promotion="strict", **kw))
