"""The Image container of darsia_amd against the reference's own classes (tests/golden/make_golden_image.py executes
image/image.py, coordinatesystem.py, basecorrection.py): origin / voxel size defaults, subregion, copy, astype, img_as,
arithmetic and comparisons, the image classes, metadata round trip, corrections on Images, transformations at construction."""
import os

import numpy as np
import pytest

import darsia_amd as da
from darsia_amd.corrections import BaseCorrection

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "image_golden.npz"))


def _im():
    return da.Image(G["im/arr"].copy(), dimensions=[3.0, 4.0], name="full")


def test_geometry_defaults():
    im = _im()
    assert np.array_equal(np.array(im.origin, dtype=float), G["im/origin"])
    assert np.array_equal(np.array(im.voxel_size), G["im/voxel_size"]) and np.array_equal(np.array(im.num_voxels), G["im/num_voxels"])
    assert np.array_equal(np.array(da.Image(G["im/arr"].copy()).origin, dtype=float), G["im/default_dims_origin"])
    assert np.array_equal(np.array(da.Image(G["im/arr"].copy(), width=2, height=1, scalar=True).dimensions, dtype=float), G["im/hw_kwargs_dims"])


@pytest.mark.parametrize("tag,roi", [("slices", (slice(1, 4), slice(2, 6))), ("voxels", "V"), ("coords", "C"), ("open", (slice(None, 2), slice(5, None)))])
def test_subregion(tag, roi):
    if roi == "V":
        roi = da.VoxelArray([[1, 2], [4, 6]])
    elif roi == "C":
        roi = da.CoordinateArray([[1.0, 2.5], [3.0, 1.0]])
    sub = _im().subregion(roi)
    assert np.array_equal(sub.img, G[f"sub_{tag}/img"])
    assert np.array_equal(np.array(sub.dimensions, dtype=float), G[f"sub_{tag}/dims"])
    assert np.array_equal(np.array(sub.origin, dtype=float), G[f"sub_{tag}/origin"])
    assert [type(sub).__name__, str(sub.name)] == list(G[f"sub_{tag}/type_name"])


def test_copy_types_and_conversions():
    im = _im()
    cp = im.copy()
    cp.img[0, 0] = 99
    assert im.img[0, 0] == float(G["copy/original_00"])
    assert str(im.astype(np.float32).img.dtype) == str(G["astype/f32_dtype"])
    s = im.astype(da.ScalarImage)
    assert [type(s).__name__, str(s.scalar), str(s.img is im.img)] == list(G["astype/class"])
    u8 = da.Image(G["img_as/u8"])
    for key, target in (("float", float), ("f32", np.float32), ("u16", np.uint16)):
        got = u8.img_as(target).img
        assert got.dtype == G[f"img_as/{key}"].dtype and np.array_equal(got, G[f"img_as/{key}"])
    assert np.array_equal(da.Image(G["img_as/float_in"]).img_as(np.uint8).img, G["img_as/float_to_u8"])


def test_arithmetic_and_comparisons():
    im = _im()
    two = im * 2
    assert np.array_equal(two.img, G["arith/mul"]) and np.array_equal((3 * im).img, G["arith/rmul"])
    assert np.array_equal((im + two).img, G["arith/add"]) and np.array_equal((two - im).img, G["arith/sub"])
    assert [type(two).__name__, type(im < 5).__name__, str((im < 5).img.dtype)] == list(G["arith/types"])
    assert np.array_equal(np.array(two.dimensions, dtype=float), G["arith/mul_dims"])
    for key, val in (("lt5", im < 5), ("ge_two", im >= two), ("eq3", im == 3), ("gt_two", im > two), ("le47", im <= 47)):
        assert np.array_equal(val.img, G[f"cmp/{key}"])
    assert str(G["arith/shape_mismatch"]) == "ValueError"
    with pytest.raises(ValueError):
        im + da.Image(np.zeros((2, 2)))


def test_classes_metadata_and_corrections():
    arr = G["im/arr"]
    sc, op = da.ScalarImage(arr.copy()), da.OpticalImage(np.zeros((4, 5, 3)))
    ex = da.ExtensiveImage(arr.copy(), dimensions=[1, 1], scalar=True)
    assert [str(sc.scalar), str(op.scalar), str(ex.scalar), type(ex.copy()).__name__] == list(G["classes/flags"])
    cp = _im().copy()
    again = da.Image(cp.img, **cp.metadata())
    assert np.array_equal([float(again.dimensions[0]), float(again.dimensions[1]), float(again.origin[0]), float(again.origin[1])], G["meta/roundtrip"])

    class Scale(BaseCorrection):
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
    assert [str(res is not base), str(res.name), str(base.name), str(float(base.img[0, 0])), str(float(res.img[0, 0]))] == list(G["corr/copy"])
    same = Scale(3.0, name="scaled")(base, overwrite=True)
    assert [str(same is base), str(base.name), str(float(base.img[0, 0]))] == list(G["corr/overwrite"])
    built = da.ScalarImage(np.ones((2, 2)), transformations=[Scale(2.0), None, Scale(5.0, name="done")])
    assert np.array_equal(built.img, G["corr/transformations"]) and str(built.name) == str(G["corr/transformations_name"])
