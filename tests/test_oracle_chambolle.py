"""Chambolle oracle vs scikit-image 0.18.3 outputs (version-labelled third-party goldens)."""
import os

import numpy as np
import pytest

from oracle import chambolle_numpy as OC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sk():
    return np.load(os.path.join(ROOT, "tests", "golden", "skimage_0_18_3_golden.npz"))


NAMES = ["c2d_f64", "c2d_f32", "c2d_fixed_f64", "c2d_fixed_f32", "crgb_f32", "crgb_fixed_f64", "c2d_one_iter", "c2d_tiny"]


@pytest.mark.parametrize("name", NAMES)
def test_chambolle_matches_skimage(sk, name):
    img = sk[name + "/img"]
    w, eps, n = sk[name + "/kw"]
    out = OC.denoise_tv_chambolle(img, weight=float(w), eps=float(eps), max_num_iter=int(n))
    ref = sk[name + "/out"]
    assert out.dtype == ref.dtype and out.shape == ref.shape
    assert np.array_equal(out, ref), f"max abs {np.abs(out - ref).max():.3e}"
    assert str(sk["meta/skimage"]) == "0.18.3"
