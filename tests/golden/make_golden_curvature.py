"""Golden vectors for CurvatureCorrection (corrections/shape/curvature.py:170-960) and BaseCorrection's dispatch
(corrections/basecorrection.py:15-95), produced by running the reference's own classes: the two files are executed
unmodified on an empty ``darsia`` shim (make_golden.py's way).

Configurations WITHOUT a "crop" step only -- cropping is ``cv2.warpPerspective`` (extract_quadrilateral_ROI) --: the
"init" / "bulge" / "stretch" coordinate maps, the cached grid, ``scipy.ndimage.map_coordinates`` gathers of orders 0 / 1,
scalar and colour images, float32 / float64 / uint8 / uint16, ``resize_factor``.  Registered as EMPTY modules so that the
imports at the top of curvature.py succeed (none of their functions is on these paths): ``cv2``, ``skimage``, and
``tomllib`` on Python < 3.11.

    python tests/golden/make_golden_curvature.py        # -> curvature_golden.npz   (system Python: NumPy 2.2, SciPy 1.15)

TEST INFRASTRUCTURE, build container only.  The output holds inputs and expected outputs (data, no source).
"""
from __future__ import annotations

import copy
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/src/darsia"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(da, rel):
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
    for name in ("cv2", "skimage", "tomllib"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:  # noqa: BLE001
                sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, HERE)
    from make_golden_image import reference_image_classes

    da = reference_image_classes()      # the reference's own Image classes (make_golden_image.py)
    Image = da.ScalarImage
    _load(da, "corrections/basecorrection.py")
    _load(da, "corrections/shape/curvature.py")

    rng = np.random.default_rng(19)
    out = {"meta/numpy": np.array(np.__version__)}
    import scipy
    out["meta/scipy"] = np.array(scipy.__version__)
    configs = {
        "bulge": {"bulge": dict(horizontal_bulge=5e-6, vertical_bulge=-3e-6)},
        "all3": {"init": dict(horizontal_bulge=4e-7, vertical_bulge=1e-7),
                 "bulge": dict(horizontal_bulge=-2e-7, vertical_bulge=3e-7, horizontal_center_offset=4),
                 "stretch": dict(horizontal_stretch=2e-7, vertical_stretch=-1e-7, vertical_center_offset=2)},
        "stretch": {"stretch": dict(horizontal_stretch=3e-6, vertical_stretch=2e-6, horizontal_center_offset=-3, vertical_center_offset=5)},
    }
    cases = [("bulge", (60, 84, 3), np.float32, 1, 1.0), ("bulge", (60, 84), np.float64, 1, 1.0), ("all3", (120, 160, 3), np.float32, 1, 1.0),
             ("all3", (120, 160), np.uint8, 1, 1.0), ("stretch", (75, 101, 3), np.uint16, 1, 1.0), ("stretch", (75, 101), np.float32, 0, 1.0),
             ("all3", (90, 120, 3), np.float64, 0, 1.0), ("bulge", (60, 84, 3), np.float32, 1, 0.5)]
    for k, (cname, shape, dtype, order, rf) in enumerate(cases):
        raw = rng.random(shape)
        if np.dtype(dtype).kind == "u":
            img = (raw * np.iinfo(dtype).max).astype(dtype)
        else:
            img = raw.astype(dtype)
        cc = da.CurvatureCorrection(config=copy.deepcopy(configs[cname]), interpolation_order=order, resize_factor=rf)
        res = cc(img)                        # BaseCorrection.__call__ on an array: corrected copy
        tag = f"k{k}"
        out[f"{tag}/img"], out[f"{tag}/out"] = img, res
        out[f"{tag}/grid"], out[f"{tag}/shape"] = np.array(cc.cache["grid"]), np.array(cc.cache["shape"])
        out[f"{tag}/spec"] = np.array(repr(dict(config=cname, order=order, resize_factor=rf)))
        assert res is not img
    # the Image branch of BaseCorrection.__call__: copy unless overwrite, metadata update from correct_metadata
    cc = da.CurvatureCorrection(config=copy.deepcopy(configs["bulge"]))
    im = Image(rng.random((40, 56)).astype(np.float32), name="frame")
    res = cc(im)
    out["image/img"], out["image/out"] = im.img, res.img
    out["image/is_new_object"] = np.array(res is not im and res.name == "frame")
    np.savez_compressed(os.path.join(HERE, "curvature_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
