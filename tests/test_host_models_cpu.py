"""Host-side mirrors: array slicing helpers and the per-pixel model / reduction classes
(NumPy semantics of the reference; no device work)."""
import numpy as np
import pytest

import darsia_amd as da
from oracle import concentration_numpy as OC


def test_array_slice_helpers():
    a = np.arange(24).reshape(2, 3, 4)
    assert np.array_equal(da.array_slice(a, 1, 0, 2), a[:, 0:2])
    assert np.array_equal(da.array_slice(a, -1, 1, None, 2), a[:, :, 1::2])
    assert da.array_slice_argument(a, 2, 0, 1) == (slice(None), slice(None), slice(0, 1, 1))
    b = np.zeros_like(a)
    b[da.array_slice_argument(a, 0, -1, None)] += 7
    assert b[-1].min() == 7 and b[0].max() == 0


def test_models_and_reduction_follow_numpy_semantics():
    rng = np.random.default_rng(0)
    x = rng.random((5, 7)).astype(np.float32)
    assert da.ScalingModel(scaling=1.0)(x) is x
    assert np.array_equal(da.ScalingModel(scaling=2.0)(x), 2.0 * x)
    assert np.array_equal(da.LinearModel(scaling=3.0, offset=0.5)(x), 3.0 * x + 0.5)
    assert np.array_equal(da.ClipModel(max_value=0.5)(x), np.clip(x, None, 0.5))
    with pytest.raises(ValueError):
        da.ClipModel()
    cm = da.CombinedModel([da.LinearModel(scaling=2.0, offset=-0.2), da.ClipModel(min_value=0.0, max_value=1.0)])
    assert np.array_equal(cm(x), np.clip(2.0 * x - 0.2, 0.0, 1.0))
    assert cm.num_parameters == 4 and isinstance(cm[1], da.ClipModel)
    rgb = rng.random((4, 6, 3)).astype(np.float32)
    for color in ("red", "green", "blue", "red+green", "gray", "negative-key"):
        assert np.array_equal(da.MonochromaticReduction(color=color)(rgb), OC.reduce(rgb, color))
    with pytest.raises(ValueError):
        da.MonochromaticReduction(color="purple")(rgb)


def test_cleaning_filter_is_the_maximum_reduced_difference():
    rng = np.random.default_rng(1)
    frames = [da.OpticalImage(rng.random((6, 8, 3)).astype(np.float32)) for _ in range(3)]
    ca = da.ConcentrationAnalysis(frames, da.MonochromaticReduction(color="green"), **{"diff option": "absolute"})
    expect = np.maximum(np.abs(frames[1].img - frames[0].img)[:, :, 1], np.abs(frames[2].img - frames[0].img)[:, :, 1])
    assert ca.threshold_cleaning_filter.dtype == np.float64
    assert np.array_equal(ca.threshold_cleaning_filter, np.maximum(0.0, expect))


def test_correction_protocol_and_transformations_at_construction():
    """basecorrection.py:15-95 and image/image.py:188-199: a correction maps arrays and Images, copies unless told to
    overwrite, updates the metadata it declares; corrections handed to the Image constructor run in place, in order."""
    import darsia_amd as da
    from darsia_amd.corrections import BaseCorrection

    class Scale(BaseCorrection):
        def __init__(self, f, name=None):
            self.f, self.new_name = f, name

        def correct_array(self, image):
            image *= self.f           # writes into its argument on purpose
            return image

        def correct_metadata(self, metadata={}):
            return {} if self.new_name is None else {"name": self.new_name}

        def save(self, path):
            pass

        def load(self, path):
            pass

    arr = np.arange(6.0).reshape(2, 3)
    twice = Scale(2.0)
    out = twice(arr)
    assert np.array_equal(out, 2 * np.arange(6.0).reshape(2, 3)) and arr[0, 1] == 1.0      # copy
    assert twice(arr, overwrite=True) is arr and arr[0, 1] == 2.0                           # in place
    im = da.ScalarImage(np.ones((2, 3)), name="raw", dimensions=[2.0, 3.0])
    res = Scale(3.0, name="scaled")(im)
    assert res is not im and res.name == "scaled" and im.name == "raw" and im.img[0, 0] == 1.0 and res.img[0, 0] == 3.0
    assert res.dimensions == [2.0, 3.0]
    same = Scale(3.0, name="scaled")(im, overwrite=True)
    assert same is im and im.name == "scaled" and im.img[0, 0] == 3.0
    # transformations at construction: applied in order, None entries skipped
    built = da.ScalarImage(np.ones((2, 2)), transformations=[Scale(2.0), None, Scale(5.0, name="done")])
    assert np.array_equal(built.img, np.full((2, 2), 10.0)) and built.name == "done"
    with pytest.raises(TypeError):
        twice([1, 2, 3])


def test_image_api_surface_subregion_types_arithmetic():
    """The thin Image container's part of the reference API (image/image.py:461-521, 665-746, 927-1058): values worked
    out by hand for a 6 x 8 image of 3 x 4 length units (voxel 0.5, origin (0, 3))."""
    import darsia_amd as da

    im = da.Image(np.arange(48.0).reshape(6, 8), dimensions=[3.0, 4.0], name="full")
    for roi in ((slice(1, 4), slice(2, 6)), da.VoxelArray([[1, 2], [4, 6]]), da.CoordinateArray([[1.0, 2.5], [3.0, 1.0]])):
        sub = im.subregion(roi)
        assert type(sub) is da.Image and np.array_equal(sub.img, im.img[1:4, 2:6]) and sub.name == "full"
        assert np.allclose(sub.dimensions, [1.5, 2.0]) and np.allclose(sub.origin, [1.0, 2.5])
        assert np.allclose(sub.coordinatesystem.coordinate([0, 0]), im.coordinatesystem.coordinate([1, 2]))
    open_ended = im.subregion((slice(None, 2), slice(5, None)))
    assert open_ended.img.shape == (2, 3) and np.allclose(open_ended.dimensions, [1.0, 1.5]) and np.allclose(open_ended.origin, [2.5, 3.0])
    with pytest.raises(ValueError):
        im.subregion([1, 2])
    # type changes
    assert im.astype(np.float32).img.dtype == np.float32 and im.img.dtype == np.float64
    as_scalar = im.astype(da.ScalarImage)
    assert isinstance(as_scalar, da.ScalarImage) and as_scalar.scalar and as_scalar.img is not im.img
    u8 = da.Image((np.arange(12).reshape(3, 4) * 20).astype(np.uint8))
    assert np.allclose(u8.img_as(float).img, u8.img / 255.0) and u8.img_as(np.float32).img.dtype == np.float32
    assert np.array_equal(da.Image(np.array([[0.5, 2.0, -3.0]])).img_as(np.uint8).img, [[128, 255, 0]])   # clipped to [-1, 1] first
    assert np.array_equal(u8.img_as(np.uint16).img, u8.img.astype(np.uint16) * 257)
    with pytest.raises(NotImplementedError):
        u8.img_as(complex)
    # arithmetic keeps type and metadata, comparisons give boolean scalar images
    two = im * 2
    assert type(two) is da.Image and two.dimensions == [3.0, 4.0] and np.array_equal(two.img, 2 * im.img)
    assert np.array_equal((3 * im).img, 3 * im.img) and np.array_equal((im + two).img, 3 * im.img)
    assert np.array_equal((two - im).img, im.img)
    with pytest.raises(ValueError):
        im + da.Image(np.zeros((2, 2)))
    with pytest.raises(ValueError):
        im * "2"
    small = im < 5
    assert isinstance(small, da.ScalarImage) and small.img.dtype == bool and small.img.sum() == 5
    assert (im >= two).img.sum() == 1 and (im == 3).img.sum() == 1 and (im > two).img.sum() == 0 and (im <= 47).img.all()
    assert isinstance(hash(im), int)
