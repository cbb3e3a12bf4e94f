"""Data-model and ROI-mask checks (integer / boolean work: exact).  The properties are those the
reference states in tests/unit/test_roi_to_mask.py and tests/unit/test_patches.py, re-expressed against
darsia_amd's classes; expected index ranges are additionally computed by hand."""
import numpy as np
import pytest

import darsia_amd as da


def _img(rows=10, cols=20):
    return da.ScalarImage(np.ones((rows, cols), np.float32), space_dim=2, dimensions=[1.0, 2.0])


def _box(c1, c2):
    return da.make_coordinate([c1, c2])


def test_default_origin_and_voxel_coordinate_round_trip():
    img = _img()
    assert np.array_equal(img.origin, [0.0, 1.0])  # ij indexing: y runs downwards from the image height
    cs = img.coordinatesystem
    assert cs.voxel_size == {"x": 0.1, "y": 0.1}
    assert np.array_equal(cs.voxel([0.55, 0.35]), [6, 5])  # row = floor((1 - .35)/.1), col = floor(.55/.1)
    np.testing.assert_allclose(cs.coordinate([6, 5]), [0.5, 0.4])
    pts = np.array([[0.05, 0.95], [1.95, 0.05]])
    assert np.array_equal(cs.voxel(pts), [[0, 0], [9, 19]])
    np.testing.assert_allclose(cs.coordinate_vector(np.array([2, 3])), [0.3, -0.2])
    assert cs.num_voxels(0.25, "x") == 3


def test_mask_type_shape_and_exact_index_range():
    img = _img()
    mask = da.roi_to_mask(_box([0.5, 0.3], [1.5, 0.7]), img)
    assert isinstance(mask, da.ScalarImage) and mask.img.dtype == np.bool_ and mask.img.shape == img.img.shape
    expected = np.zeros((10, 20), bool)
    # rows floor((1-.7)/.1) = floor(3.0000000000000004) = 3 .. floor((1-.3)/.1) = floor(6.999999999999999) = 6
    # (the reference's float arithmetic, kept), cols floor(.5/.1) = 5 .. floor(1.5/.1) = 15
    expected[3:6, 5:15] = True
    assert np.array_equal(mask.img, expected)
    v = img.coordinatesystem.voxel(_box([0.5, 0.3], [1.5, 0.7]))
    assert mask.img.sum() == (v[:, 0].max() - v[:, 0].min()) * (v[:, 1].max() - v[:, 1].min())


def test_full_partial_and_outside_rois():
    img = _img(8, 16)
    cs = img.coordinatesystem
    full = da.roi_to_mask(_box(cs._coordinate_of_origin_voxel.tolist(), cs._coordinate_of_opposite_voxel.tolist()), img)
    assert full.img.all()
    part = da.roi_to_mask(_box([-1.0, -0.5], [3.0, 1.5]), img)
    assert part.img.shape == (8, 16) and part.img.any()
    assert not da.roi_to_mask(_box([0.0, 1.1], [2.0, 1.5]), img).img.any()


def test_list_is_union_and_voxel_rois():
    img = _img()
    a, b = _box([0.0, 0.3], [1.5, 0.7]), _box([0.5, 0.3], [2.0, 0.7])
    ma, mb = da.roi_to_mask(a, img).img, da.roi_to_mask(b, img).img
    assert np.array_equal(da.roi_to_mask([a, b], img).img, ma | mb)
    assert np.array_equal(da.roi_to_mask([a], img).img, ma)
    empty = da.roi_to_mask([], img)
    assert isinstance(empty, da.ScalarImage) and not empty.img.any()
    vox = da.make_voxel([[2, 3], [5, 9]])
    m = da.roi_to_mask(vox, img).img
    assert m.sum() == 3 * 6 and m[2:5, 3:9].all()
    with pytest.raises(TypeError):
        da.roi_to_mask(np.array([[0, 0], [1, 1]]), img)


def test_patches_partition_the_image():
    arr = np.arange(1, 17).reshape(2, 8)
    patches = da.Patches(da.Image(arr, space_dim=2), num_patches=(2, 4))
    got = [[patches(i, j).img.tolist() for j in range(4)] for i in range(2)]
    assert got == [[[[1, 2]], [[3, 4]], [[5, 6]], [[7, 8]]], [[[9, 10]], [[11, 12]], [[13, 14]], [[15, 16]]]]
    # "Check that assemble works" (tests/unit/test_patches.py:27-28), also with overlapping patches
    assert np.array_equal(patches.assemble().img, arr)
    big = np.arange(100 * 130).reshape(100, 130)
    over = da.Patches(da.Image(big, dimensions=[1.0, 1.3]), num_patches=[3, 4], rel_overlap=0.1)
    assert np.array_equal(over.assemble().img, big)


def test_indexing_helpers():
    assert da.interpret_indexing("y", "ij") == (0, True) and da.interpret_indexing("x", "ij") == (1, False)
    assert da.to_cartesian_indexing(0, "ij") == "y" and da.to_matrix_indexing("x", "xy") == "j"
    assert isinstance(da.make_voxel([1.7, 2.2]), da.Voxel) and da.make_voxel([1.7, 2.2]).tolist() == [1, 2]
    assert da.make_voxel([[1, 2]], matrix_indexing=False).tolist() == [[2, 1]]
    with pytest.raises(ValueError):
        da.interpret_indexing("q", "ij")
