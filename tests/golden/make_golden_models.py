"""Golden vectors for the NumPy-only operators around the TV path, produced by importing the reference's own
modules (file by file onto an empty ``darsia`` shim, like make_golden.py):

    corrections/color/colorbalance.py   AffineBalance.apply_balance / find_balance  (:140-196)
    signals/models/basemodel.py         Model
    signals/models/clipmodel.py         ClipModel.__call__                          (:87-106)
    signals/models/combinedmodel.py     CombinedModel.__call__ (argument-count dispatch, :29-52)

and, under /opt/conda's Python 3.9 (which has the real scikit-image 0.18.3 that utils/dtype.py calls into):

    utils/dtype.py                      convert_dtype                               (:9-30)

TEST INFRASTRUCTURE; run in the build container only:

    python tests/golden/make_golden_models.py            # models_golden.npz
    /opt/conda/bin/python3.9 tests/golden/make_golden_models.py dtype   # dtype_golden.npz

The output holds inputs and expected outputs only (data, no source).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/src/darsia"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(da, rel):
    name = "darsia_ref_" + rel.replace("/", "_")[:-3]
    spec = importlib.util.spec_from_file_location(name, f"{REF}/{rel}")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for k, v in vars(m).items():
        if not k.startswith("_"):
            setattr(da, k, v)
    return m


def models():
    da = types.ModuleType("darsia")
    sys.modules["darsia"] = da

    class Image:  # stand-in for the isinstance checks of the models
        def __init__(self, img):
            self.img = img

        def copy(self):
            return Image(self.img.copy())

    da.Image = Image
    for rel in ("corrections/color/colorbalance.py", "signals/models/basemodel.py", "signals/models/clipmodel.py",
                "signals/models/combinedmodel.py"):
        _load(da, rel)
    out = {"meta/numpy": np.array(np.__version__)}
    rng = np.random.default_rng(2024)

    # ---- AffineBalance.apply_balance: img @ A + b on (32, 48, 3), float64 (after img_as_float) and float32
    A = np.eye(3) + 0.05 * rng.normal(size=(3, 3))
    b = 0.01 * rng.normal(size=3)
    bal = da.AffineBalance()
    bal.balance_scaling, bal.balance_translation = A, b
    for tag, dt in (("f64", np.float64), ("f32", np.float32)):
        img = rng.random((32, 48, 3)).astype(dt)
        out[f"balance_{tag}/img"] = img
        out[f"balance_{tag}/out"] = bal.apply_balance(img)
    out["balance/A"], out["balance/b"] = A, b
    u8 = (rng.random((32, 48, 3)) * 255).astype(np.uint8)
    out["balance_u8/img"] = u8
    out["balance_u8/out"] = bal.apply_balance(u8 / 255.0)   # the caller's img_as_float (colorcorrection.py:352-388)
    # find_balance on noisy swatches of a known affine map (Powell, tol 1e-6)
    src = rng.random((24, 3))
    dst = src @ A + b + 1e-3 * rng.normal(size=(24, 3))
    fit = da.AffineBalance()
    fit.find_balance(src, dst)
    out["fit/src"], out["fit/dst"] = src, dst
    out["fit/A"], out["fit/b"] = fit.balance_scaling, fit.balance_translation

    # ---- ClipModel
    sig = rng.normal(0.3, 0.5, size=(40, 56)).astype(np.float32)
    out["clip/img"] = sig
    out["clip/both"] = da.ClipModel(min_value=0.0, max_value=1.0)(sig)
    out["clip/lower"] = da.ClipModel(min_value=0.1)(sig)
    out["clip/upper"] = da.ClipModel(max_value=0.45)(sig)
    out["clip/keyed"] = da.ClipModel(key="model", model_min_value=-0.2, model_max_value=0.6)(sig)
    sig64 = sig.astype(np.float64)
    out["clip/both_f64"] = da.ClipModel(min_value=0.0, max_value=1.0)(sig64)
    try:
        da.ClipModel()
        out["clip/empty_raises"] = np.array("no")
    except ValueError:
        out["clip/empty_raises"] = np.array("ValueError")

    # ---- CombinedModel: one-argument models chained; a two-argument model gets the extra positional argument
    class Scale(da.Model):
        def __init__(self, s):
            self.s = s

        def __call__(self, img):
            return self.s * img

        def update_model_parameters(self, *a, **k):
            pass

        def calibrate(self, *a, **k):
            return True

    class AddArg(da.Model):
        def __call__(self, img, extra):
            return img + extra

        def update_model_parameters(self, *a, **k):
            pass

        def calibrate(self, *a, **k):
            return True

    try:
        comb = da.CombinedModel([Scale(1.5), da.ClipModel(min_value=0.0, max_value=1.0), Scale(0.5)])
        out["combined/chain"] = comb(sig)
        comb2 = da.CombinedModel([Scale(2.0), AddArg(), da.ClipModel(max_value=1.0)])
        out["combined/with_arg"] = comb2(sig, np.float32(0.25))
        out["combined/num_parameters"] = np.array(comb.num_parameters)
    except TypeError as e:  # abstract methods of the reference's Model differ: record it instead of guessing
        out["combined/error"] = np.array(str(e))
    np.savez_compressed(os.path.join(HERE, "models_golden.npz"), **out)
    print("wrote models_golden.npz:", len(out), "arrays")


def dtype():
    import skimage

    da = types.ModuleType("darsia")
    sys.modules["darsia"] = da
    m = _load(da, "utils/dtype.py")
    rng = np.random.default_rng(77)
    out = {"meta/skimage": np.array(skimage.__version__), "meta/numpy": np.array(np.__version__)}
    srcs = {
        "f32": rng.random((12, 17)).astype(np.float32),
        "f64": rng.random((12, 17, 3)),
        "f64_signed": rng.random((9, 11)) * 2.0 - 1.0,
        "u8": (rng.random((12, 17, 3)) * 255).astype(np.uint8),
        "u16": (rng.random((12, 17)) * 65535).astype(np.uint16),
    }
    import warnings

    for sname, img in srcs.items():
        out[f"src/{sname}"] = img
        for tname, dt in (("u8", np.uint8), ("u16", np.uint16), ("f32", np.float32), ("f64", np.float64)):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out[f"{sname}_to_{tname}"] = m.convert_dtype(img, dt)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        same = m.convert_dtype(srcs["f32"], np.int32)
        out["unsupported/warned"] = np.array(len(w))
        out["unsupported/returned_same_object"] = np.array(same is srcs["f32"])
    np.savez_compressed(os.path.join(HERE, "dtype_golden.npz"), **out)
    print("wrote dtype_golden.npz:", len(out), "arrays")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dtype":
        dtype()
    else:
        models()
