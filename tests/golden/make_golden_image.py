"""Golden vectors for the Image container's API surface (image/image.py: construction and metadata :35-200, copy :363-370,
update_metadata :441-454, astype / img_as :461-521, metadata :525-546, subregion :665-746, arithmetic and comparisons
:927-1058; ScalarImage :1860, ExtensiveImage :2064, OpticalImage :2081) and for BaseCorrection's dispatch on real Images
(corrections/basecorrection.py:15-95), produced by the reference's own classes.

image/image.py is executed unmodified on an empty ``darsia`` shim together with utils/point.py, image/indexing.py and
image/coordinatesystem.py.  Its top-level imports of ``cv2`` and ``plotly`` (plotting / I-O, not exercised) are satisfied
by EMPTY modules of those names.  scikit-image must be real (``img_as``): run under /opt/conda's Python 3.9:

    /opt/conda/bin/python3.9 tests/golden/make_golden_image.py      # -> image_golden.npz

TEST INFRASTRUCTURE, build container only.  The output holds inputs and expected outputs (data, no source).
"""
from __future__ import annotations

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


def reference_image_classes():
    """The ``darsia`` shim with the reference's point / indexing / coordinate-system / image modules loaded."""
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    for n in ("plotly", "plotly.express", "plotly.graph_objects", "plotly.subplots"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["plotly.subplots"].make_subplots = None
    sys.modules["plotly"].express = sys.modules["plotly.express"]
    sys.modules["plotly"].graph_objects = sys.modules["plotly.graph_objects"]
    da = types.ModuleType("darsia")
    sys.modules["darsia"] = da
    for rel in ("utils/point.py", "image/indexing.py", "image/coordinatesystem.py", "image/image.py", "utils/standard_images.py"):
        _load(da, rel)
    return da


def main():
    da = reference_image_classes()
    _load(da, "corrections/basecorrection.py")
    rng = np.random.default_rng(3)
    out = {"meta/numpy": np.array(np.__version__)}
    arr = np.arange(48.0).reshape(6, 8)
    im = da.Image(arr.copy(), dimensions=[3.0, 4.0], name="full")
    out["im/arr"] = arr
    out["im/origin"], out["im/voxel_size"], out["im/num_voxels"] = np.array(im.origin, dtype=float), np.array(im.voxel_size), np.array(im.num_voxels)
    out["im/default_dims_origin"] = np.array(da.Image(arr.copy()).origin, dtype=float)
    out["im/hw_kwargs_dims"] = np.array(da.Image(arr.copy(), width=2, height=1, scalar=True).dimensions, dtype=float)
    # subregion by slices, VoxelArray, CoordinateArray, open-ended slices
    for tag, roi in (("slices", (slice(1, 4), slice(2, 6))), ("voxels", da.VoxelArray([[1, 2], [4, 6]])),
                     ("coords", da.CoordinateArray([[1.0, 2.5], [3.0, 1.0]])), ("open", (slice(None, 2), slice(5, None)))):
        sub = im.subregion(roi)
        out[f"sub_{tag}/img"], out[f"sub_{tag}/dims"], out[f"sub_{tag}/origin"] = sub.img, np.array(sub.dimensions, dtype=float), np.array(sub.origin, dtype=float)
        out[f"sub_{tag}/type_name"] = np.array([type(sub).__name__, str(sub.name)])
    # copy is deep
    cp = im.copy()
    cp.img[0, 0] = 99
    out["copy/original_00"] = np.array(im.img[0, 0])
    # astype / img_as
    out["astype/f32_dtype"] = np.array(str(im.astype(np.float32).img.dtype))
    as_scalar = im.astype(da.ScalarImage)
    out["astype/class"] = np.array([type(as_scalar).__name__, str(as_scalar.scalar), str(as_scalar.img is im.img)])
    u8 = da.Image((np.arange(12).reshape(3, 4) * 20).astype(np.uint8))
    out["img_as/u8"] = u8.img
    out["img_as/float"], out["img_as/f32"], out["img_as/u16"] = u8.img_as(float).img, u8.img_as(np.float32).img, u8.img_as(np.uint16).img
    fl = da.Image(np.array([[0.5, 2.0, -3.0]]))
    out["img_as/float_in"], out["img_as/float_to_u8"] = fl.img, fl.img_as(np.uint8).img
    # arithmetic / comparisons
    two = im * 2
    out["arith/mul"], out["arith/rmul"], out["arith/add"], out["arith/sub"] = two.img, (3 * im).img, (im + two).img, (two - im).img
    out["arith/types"] = np.array([type(two).__name__, type(im < 5).__name__, str((im < 5).img.dtype)])
    out["arith/mul_dims"] = np.array(two.dimensions, dtype=float)
    out["cmp/lt5"], out["cmp/ge_two"], out["cmp/eq3"], out["cmp/gt_two"], out["cmp/le47"] = (im < 5).img, (im >= two).img, (im == 3).img, (im > two).img, (im <= 47).img
    try:
        im + da.Image(np.zeros((2, 2)))
        out["arith/shape_mismatch"] = np.array("no error")
    except ValueError:
        out["arith/shape_mismatch"] = np.array("ValueError")
    # image classes
    sc = da.ScalarImage(arr.copy())
    op = da.OpticalImage(rng.random((4, 5, 3)))
    ex = da.ExtensiveImage(arr.copy(), dimensions=[1, 1], scalar=True)
    out["classes/flags"] = np.array([str(sc.scalar), str(op.scalar), str(ex.scalar), type(ex.copy()).__name__])
    # metadata round trip keeps the spatial description
    again = da.Image(cp.img, **cp.metadata())
    out["meta/roundtrip"] = np.array([float(again.dimensions[0]), float(again.dimensions[1]), float(again.origin[0]), float(again.origin[1])])
    # BaseCorrection on real Images + transformations at construction

    class Scale(da.BaseCorrection):
        def __init__(self, f, name=None):
            self.f, self.new_name = f, name

        def correct_array(self, image):
            image *= self.f
            return image

        def correct_metadata(self, metadata={}):
            return {} if self.new_name is None else {"name": self.new_name}

        def save(self, path):
            pass

        def load(self, path):
            pass

    base = da.ScalarImage(np.ones((2, 3)), name="raw", dimensions=[2.0, 3.0])
    res = Scale(3.0, name="scaled")(base)
    out["corr/copy"] = np.array([str(res is not base), str(res.name), str(base.name), str(float(base.img[0, 0])), str(float(res.img[0, 0]))])
    same = Scale(3.0, name="scaled")(base, overwrite=True)
    out["corr/overwrite"] = np.array([str(same is base), str(base.name), str(float(base.img[0, 0]))])
    built = da.ScalarImage(np.ones((2, 2)), transformations=[Scale(2.0), None, Scale(5.0, name="done")])
    out["corr/transformations"] = built.img
    out["corr/transformations_name"] = np.array(str(built.name))
    np.savez_compressed(os.path.join(HERE, "image_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
