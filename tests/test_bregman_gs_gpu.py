"""scikit-image's Gauss-Seidel split Bregman on the device (wavefront sweep, through the C ABI) against
scikit-image 0.18.3 outputs and the NumPy oracle: bit-exact."""
import os

import numpy as np
import pytest

import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
from oracle import bregman_gs_numpy as bg

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_0_18_3_golden.npz"))
CASES = ["b_aniso_f64", "b_iso_f64", "b_aniso_f32", "b_iso_f32", "b_rgb_aniso_f32", "b_rgb_iso_f64", "b_wide_aniso_f64",
         "b_tall_iso_f32", "b_uint8"]


@pytest.mark.parametrize("name", CASES)
def test_golden_bit_exact(name):
    w, eps, it, iso = G[name + "/kw"]
    out = da.denoise_tv_bregman(G[name + "/img"], weight=w, max_num_iter=int(it), eps=eps, isotropic=bool(iso))
    assert out.dtype == G[name + "/out"].dtype and out.shape == G[name + "/out"].shape
    np.testing.assert_array_equal(out, G[name + "/out"])


@pytest.mark.parametrize("shape,dtype,iso", [((150, 211, 3), np.float32, False), ((97, 300), np.float64, True),
                                             ((1, 40), np.float64, False), ((33, 1), np.float32, True)])
def test_vs_oracle_other_shapes_bit_exact_with_iteration_count(shape, dtype, iso):
    img = synthetic_image(shape[0], shape[1], 3).astype(dtype)
    img = np.ascontiguousarray(img if len(shape) == 3 else img[..., 0])
    out, n = da.denoise_tv_bregman(img, weight=1.5, max_num_iter=25, eps=2e-3, isotropic=iso, return_iters=True)
    ref, n_ref = bg.denoise_tv_bregman(img, weight=1.5, max_num_iter=25, eps=2e-3, isotropic=iso, return_iterations=True)
    assert n == n_ref
    np.testing.assert_array_equal(out, ref)


def test_tvd_methods_dispatch_to_it():
    img = G["b_rgb_aniso_f32/img"]
    w, eps, it, _ = G["b_rgb_aniso_f32/kw"]
    out = da.TVD(method="anisotropic bregman", weight=w, max_num_iter=int(it), eps=eps)(img)
    np.testing.assert_array_equal(out, G["b_rgb_aniso_f32/out"])
    img = G["b_iso_f64/img"]
    w, eps, it, _ = G["b_iso_f64/kw"]
    out = da.tvd(img, method="isotropic bregman", weight=w, max_num_iter=int(it), eps=eps)
    np.testing.assert_array_equal(out, G["b_iso_f64/out"])


def test_channel_axis_runs_channels_separately():
    img = G["b_rgb_aniso_f32/img"]
    out = da.denoise_tv_bregman(img, weight=2.0, max_num_iter=30, eps=1e-3, isotropic=False, channel_axis=-1)
    for c in range(3):
        ref = bg.denoise_tv_bregman(np.ascontiguousarray(img[..., c]), weight=2.0, max_num_iter=30, eps=1e-3, isotropic=False)
        np.testing.assert_array_equal(out[..., c], ref)


def test_medium_image_properties():
    """1000 x 1500 x 3: constants are fixed points and the result is deterministic across runs."""
    const = np.full((64, 80, 3), 0.37, np.float32)
    np.testing.assert_array_equal(da.denoise_tv_bregman(const, weight=1.0, max_num_iter=5, eps=0.0), const)
    img = synthetic_image(1000, 1500, 3)
    a = da.denoise_tv_bregman(img, weight=1.0, max_num_iter=4, eps=0.0, isotropic=False)
    b = da.denoise_tv_bregman(img, weight=1.0, max_num_iter=4, eps=0.0, isotropic=False)
    np.testing.assert_array_equal(a, b)
    assert np.isfinite(a).all() and abs(float(a.mean()) - float(img.mean())) < 1e-2


def test_small_random_sweep_chambolle_and_gauss_seidel():
    """A short run of tests/fuzz/fuzz_others.py's generator inside the suite (bit-exact incl. iteration counts)."""
    from oracle import chambolle_numpy as CH
    rng = np.random.default_rng(77)
    for _ in range(30):
        h, w = int(rng.integers(1, 60)), int(rng.integers(1, 300))
        c = int(rng.choice([0, 1, 3]))
        shape = (h, w) if c == 0 else (h, w, c)
        dt = rng.choice([np.float32, np.float64])
        img = rng.random(shape).astype(dt)
        it = int(rng.integers(1, 12))
        eps = float(rng.choice([0.0, 1e-3, 1e-2]))
        wgt = float(rng.choice([0.05, 0.2, 1.0]))
        o, n = da.denoise_tv_chambolle(img, weight=wgt, eps=eps, max_num_iter=it, return_iters=True)
        r, nr = CH.denoise_tv_chambolle(img, weight=wgt, eps=eps, max_num_iter=it, return_iters=True)
        assert n == nr
        np.testing.assert_array_equal(o, r)
        iso = bool(rng.random() < 0.5)
        o, n = da.denoise_tv_bregman(img, weight=wgt * 5, max_num_iter=it, eps=eps, isotropic=iso, return_iters=True)
        r, nr = bg.denoise_tv_bregman(img, weight=wgt * 5, max_num_iter=it, eps=eps, isotropic=iso, return_iterations=True)
        assert n == nr
        np.testing.assert_array_equal(o, r)
