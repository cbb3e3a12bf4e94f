"""Golden vectors for ConcentrationAnalysis (multi_image_analysis/concentrationanalysis.py:20-437), produced by running
the reference's own class: its module and the modules it drives are loaded file by file onto an empty ``darsia`` shim
(like make_golden.py / make_golden_models.py):

    signals/reduction/signalreduction.py, monochromatic.py      MonochromaticReduction (:28-95)
    signals/models/basemodel.py, linearmodel.py, clipmodel.py, combinedmodel.py
    utils/array_slice.py, dtype.py, derivatives.py, utils/linear_solvers/{solver,jacobi}.py
    restoration/split_bregman_tvd.py, restoration/tvd.py        TVD("heterogeneous bregman"), fresh Jacobi per call
    multi_image_analysis/concentrationanalysis.py               ConcentrationAnalysis

Two things stand in for what the container lacks, neither on the exercised paths' arithmetic:
  * ``cv2`` is registered as an EMPTY module so that ``import cv2`` at the top of those files succeeds; no cv2 function is
    called (the "gray" reduction, which is cv2.cvtColor, is not part of these vectors);
  * ``numba.njit`` is an identity decorator if numba does not import (isotropic shrinkage is not exercised);
  * (the images are the reference's own ``OpticalImage`` / ``ScalarImage``, loaded as make_golden_image.py does).
scikit-image must be real (``skimage.util.compare_images`` is the "absolute" difference, ``img_as_float`` converts integer
frames): run under /opt/conda's Python 3.9 (scikit-image 0.18.3, NumPy 1.26):

    /opt/conda/bin/python3.9 tests/golden/make_golden_concentration.py      # -> concentration_golden.npz

TEST INFRASTRUCTURE, build container only.  The output holds inputs and expected outputs (data, no source).
"""
from __future__ import annotations

import copy
import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/src/darsia"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(da, rel):
    """Execute one reference file as a module.  The source runs unmodified; it is compiled with the ``annotations`` future
    flag (annotations stay unevaluated), which Python 3.9 needs for the ``X | Y`` type hints some of the files use."""
    import __future__

    name = "darsia_ref_" + rel.replace("/", "_")[:-3]
    path = f"{REF}/{rel}"
    m = types.ModuleType(name)
    m.__file__ = path
    sys.modules[name] = m
    with open(path) as f:
        code = compile(f.read(), path, "exec", flags=__future__.annotations.compiler_flag, dont_inherit=True)
    exec(code, m.__dict__)
    for k, v in vars(m).items():
        if not k.startswith("_"):
            setattr(da, k, v)
    return m


def main():
    import skimage

    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    try:
        import numba  # noqa: F401
    except Exception:  # noqa: BLE001 -- broken in /opt/conda
        nb = types.ModuleType("numba")
        nb.njit = lambda *a, **k: (lambda f: f)
        sys.modules["numba"] = nb
    sys.path.insert(0, HERE)
    from make_golden_image import reference_image_classes

    # the reference's own Image classes (image/image.py executed with empty ``cv2`` / ``plotly`` modules, make_golden_image.py)
    da = reference_image_classes()
    Image, ScalarImage = da.OpticalImage, da.ScalarImage
    for rel in ("utils/array_slice.py", "utils/dtype.py", "utils/derivatives.py", "utils/linear_solvers/solver.py",
                "utils/linear_solvers/jacobi.py", "restoration/split_bregman_tvd.py", "restoration/tvd.py",
                "signals/reduction/signalreduction.py", "signals/reduction/monochromatic.py",
                "signals/models/basemodel.py", "signals/models/linearmodel.py", "signals/models/clipmodel.py",
                "signals/models/combinedmodel.py", "multi_image_analysis/concentrationanalysis.py"):
        _load(da, rel)

    rng = np.random.default_rng(77)
    h, w = 36, 52
    out = {"meta/numpy": np.array(np.__version__), "meta/skimage": np.array(skimage.__version__)}

    def frames(dtype):
        base = (rng.random((h, w, 3)) * 0.3 + 0.3).astype(dtype)
        yy, xx = np.mgrid[0:h, 0:w]
        blob = np.exp(-(((yy - h / 2) / (h / 5)) ** 2 + ((xx - w / 3) / (w / 6)) ** 2))
        probe = (base + 0.3 * blob[..., None] * np.array([1.0, 0.4, -0.2]) + rng.normal(0, 0.02, base.shape)).astype(dtype)
        return base, probe

    cases = []
    # (tag, dtype, diff option, colour, extra baselines for the cleaning filter, model spec, restoration spec, restoration first)
    for k, (opt, color) in enumerate([("absolute", "red"), ("positive", "green"), ("negative", "blue"), ("plain", "red+green"),
                                      ("absolute", "negative-key"), ("positive", "red")]):
        cases.append((f"c{k}", np.float32 if k % 2 == 0 else np.float64, opt, color, 0, "linear_clip", None, False))
    cases.append(("clean32", np.float32, "positive", "red", 2, "linear_clip", None, False))
    cases.append(("clean64", np.float64, "absolute", "green", 1, "clip_only", None, False))
    cases.append(("tvd64", np.float64, "positive", "red", 0, "linear_clip", dict(weight=0.05, omega=1.0, max_num_iter=7), False))
    cases.append(("tvd64_first", np.float64, "absolute", "blue", 0, "linear_clip", dict(weight=0.08, omega=0.9, max_num_iter=6), True))
    cases.append(("tvd32", np.float32, "positive", "green", 0, "scaling", dict(weight=0.05, omega=1.0, max_num_iter=7), False))
    cases.append(("tvd64_clean", np.float64, "positive", "red", 1, "linear_clip", dict(weight=0.06, omega=1.0, max_num_iter=5), True))

    for tag, dtype, opt, color, nclean, model_spec, rest_spec, rest_first in cases:
        base, probe = frames(dtype)
        extra = [(base + rng.normal(0, 0.01, base.shape)).astype(dtype) for _ in range(nclean)]
        if model_spec == "linear_clip":
            model = da.CombinedModel([da.LinearModel(scaling=2.5, offset=-0.05), da.ClipModel(min_value=0.0, max_value=1.0)])
        elif model_spec == "clip_only":
            model = da.ClipModel(min_value=0.02)
        else:
            model = da.ScalingModel(scaling=3.0)
        restoration = None
        if rest_spec is not None:
            # a fresh Jacobi per operator: the reference's def-time default solver caches its diagonal
            restoration = da.TVD(method="heterogeneous bregman", eps=None, solver=da.Jacobi(), **rest_spec)
        ca = da.ConcentrationAnalysis([Image(base)] + [Image(e) for e in extra] if nclean else Image(base),
                                      da.MonochromaticReduction(color=color), None, restoration, model,
                                      **{"diff option": opt, "restoration -> model": rest_first})
        res = ca(Image(probe, name="frame"))
        assert isinstance(res, ScalarImage) and res.name == "frame"
        out[f"{tag}/base"], out[f"{tag}/probe"] = base, probe
        for i, e in enumerate(extra):
            out[f"{tag}/extra{i}"] = e
        out[f"{tag}/out"] = res.img
        if ca.threshold_cleaning_filter is not None:
            out[f"{tag}/filter"] = ca.threshold_cleaning_filter
        out[f"{tag}/spec"] = np.array(repr(dict(diff=opt, color=color, nclean=nclean, model=model_spec, rest=rest_spec,
                                                rest_first=rest_first)))
    # an integer frame: converted with img_as(float) and a warning (concentrationanalysis.py:241-246)
    base, probe = frames(np.float64)
    u8 = (np.clip(probe, 0, 1) * 255).astype(np.uint8)
    ca = da.ConcentrationAnalysis(Image(base), da.MonochromaticReduction(color="red"), None, None, None, **{"diff option": "plain"})
    import warnings
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        res = ca(Image(u8))
    out["u8/base"], out["u8/probe"], out["u8/out"] = base, u8, res.img
    out["u8/warned"] = np.array(any("converted to float" in str(x.message) for x in wlist))
    np.savez_compressed(os.path.join(HERE, "concentration_golden.npz"), **out)
    print("wrote", len(out), "arrays;", {k: v.dtype.name for k, v in out.items() if k.endswith("/out")})


if __name__ == "__main__":
    main()
