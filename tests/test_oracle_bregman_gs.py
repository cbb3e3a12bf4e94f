"""The restatement of scikit-image's Gauss-Seidel split Bregman against scikit-image 0.18.3 outputs."""
import os

import numpy as np
import pytest

from oracle import bregman_gs_numpy as bg

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_0_18_3_golden.npz"))
CASES = ["b_aniso_f64", "b_iso_f64", "b_aniso_f32", "b_iso_f32", "b_rgb_aniso_f32", "b_rgb_iso_f64", "b_wide_aniso_f64",
         "b_tall_iso_f32", "b_uint8"]


@pytest.mark.parametrize("name", CASES)
def test_wavefront_oracle_matches_skimage_bit_exact(name):
    w, eps, it, iso = G[name + "/kw"]
    out = bg.denoise_tv_bregman(G[name + "/img"], weight=w, max_num_iter=int(it), eps=eps, isotropic=bool(iso))
    assert out.dtype == G[name + "/out"].dtype and out.shape == G[name + "/out"].shape
    np.testing.assert_array_equal(out, G[name + "/out"])


@pytest.mark.parametrize("name", ["b_wide_aniso_f64", "b_tall_iso_f32"])
def test_raster_and_wavefront_orders_give_identical_bits(name):
    w, eps, it, iso = G[name + "/kw"]
    a = bg.denoise_tv_bregman(G[name + "/img"], weight=w, max_num_iter=int(it), eps=eps, isotropic=bool(iso), sweep="raster")
    np.testing.assert_array_equal(a, G[name + "/out"])
