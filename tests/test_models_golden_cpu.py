"""Host-side operators against goldens produced by importing the reference's own modules
(tests/golden/make_golden_models.py): convert_dtype (utils/dtype.py:9-30, real scikit-image 0.18.3 behind it),
ClipModel (signals/models/clipmodel.py:87-106), CombinedModel's argument-count dispatch
(signals/models/combinedmodel.py:29-52), AffineBalance.find_balance (corrections/color/colorbalance.py:151-184).
No GPU involved."""
import os
import warnings

import numpy as np
import pytest

import darsia_amd as da

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def models():
    return np.load(os.path.join(HERE, "golden", "models_golden.npz"))


@pytest.fixture(scope="module")
def dtypes():
    return np.load(os.path.join(HERE, "golden", "dtype_golden.npz"))


DT = {"u8": np.uint8, "u16": np.uint16, "f32": np.float32, "f64": np.float64}


@pytest.mark.parametrize("src", ["f32", "f64", "f64_signed", "u8", "u16"])
@pytest.mark.parametrize("dst", ["u8", "u16", "f32", "f64"])
def test_convert_dtype_matches_the_reference(dtypes, src, dst):
    img = dtypes[f"src/{src}"]
    want = dtypes[f"{src}_to_{dst}"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = da.convert_dtype(img, DT[dst])
    assert got.dtype == want.dtype and got.shape == want.shape
    assert np.array_equal(got, want), f"{src} -> {dst}: max abs {np.abs(got.astype(float) - want.astype(float)).max()}"


def test_convert_dtype_unsupported_target_warns_and_returns_the_input(dtypes):
    img = dtypes["src/f32"]
    with pytest.warns(UserWarning):
        out = da.convert_dtype(img, np.int32)
    assert out is img
    assert int(dtypes["unsupported/warned"]) == 1 and bool(dtypes["unsupported/returned_same_object"])


def test_clip_model(models):
    sig = models["clip/img"]
    for key, m in (("both", da.ClipModel(min_value=0.0, max_value=1.0)), ("lower", da.ClipModel(min_value=0.1)),
                   ("upper", da.ClipModel(max_value=0.45)),
                   ("keyed", da.ClipModel(key="model", model_min_value=-0.2, model_max_value=0.6))):
        out = m(sig)
        assert out.dtype == models[f"clip/{key}"].dtype and np.array_equal(out, models[f"clip/{key}"])
    assert np.array_equal(da.ClipModel(min_value=0.0, max_value=1.0)(sig.astype(np.float64)), models["clip/both_f64"])
    assert str(models["clip/empty_raises"]) == "ValueError"
    with pytest.raises(ValueError):
        da.ClipModel()
    img = da.Image(sig.copy())
    res = da.ClipModel(min_value=0.0, max_value=1.0)(img)
    assert res is not img and np.array_equal(res.img, models["clip/both"]) and np.array_equal(img.img, sig)
    with pytest.raises(ValueError):
        da.ClipModel(min_value=0.0)(3.0)


def test_combined_model_dispatch(models):
    sig = models["clip/img"]

    class Scale:
        def __init__(self, s):
            self.s = s

        def __call__(self, img):
            return self.s * img

    class AddArg:
        def __call__(self, img, extra):
            return img + extra

    comb = da.CombinedModel([Scale(1.5), da.ClipModel(min_value=0.0, max_value=1.0), Scale(0.5)])
    out = comb(sig)
    assert out.dtype == models["combined/chain"].dtype and np.array_equal(out, models["combined/chain"])
    assert comb.num_parameters == int(models["combined/num_parameters"])
    comb2 = da.CombinedModel([Scale(2.0), AddArg(), da.ClipModel(max_value=1.0)])
    out2 = comb2(sig, np.float32(0.25))
    assert out2.dtype == models["combined/with_arg"].dtype and np.array_equal(out2, models["combined/with_arg"])
    assert np.array_equal(sig, models["clip/img"])  # input untouched (result = img.copy())


def test_affine_balance_fit(models):
    bal = da.AffineBalance()
    bal.find_balance(models["fit/src"], models["fit/dst"])
    np.testing.assert_allclose(bal.balance_scaling, models["fit/A"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(bal.balance_translation, models["fit/b"], rtol=1e-9, atol=1e-12)
