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


def test_unit_coordinate_system_known_answers():
    """The known-answer values of the reference's tests/unit/test_coordinatesystem.py (a 3 x 4 image of unit dimensions):
    attributes, voxel / coordinate lists, length, num_voxels, coordinate, voxel, coordinate_vector, for every input
    container the reference's test feeds in."""
    cs = da.Image(np.ones((3, 4), dtype=float), dimensions=[1, 1], space_dim=2).coordinatesystem
    assert cs.dim == 2 and cs.shape == (3, 4) and cs.axes == "xy" and np.allclose(cs.dimensions, [1, 1])
    assert np.allclose([cs.voxel_size["x"], cs.voxel_size["y"]], [1 / 4, 1 / 3])
    assert np.allclose(cs._coordinate_of_origin_voxel, [0, 1]) and np.allclose(cs._coordinate_of_opposite_voxel, [1, 0])
    assert np.allclose([cs.domain[k] for k in ("xmin", "xmax", "ymin", "ymax")], [0, 1, 0, 1])
    assert np.allclose(cs.voxels, [[0, 0], [1, 0], [2, 0], [0, 1], [1, 1], [2, 1], [0, 2], [1, 2], [2, 2], [0, 3], [1, 3], [2, 3]])
    assert np.allclose(cs.coordinates, [[0, 1], [0, 2 / 3], [0, 1 / 3], [1 / 4, 1], [1 / 4, 2 / 3], [1 / 4, 1 / 3],
                                        [2 / 4, 1], [2 / 4, 2 / 3], [2 / 4, 1 / 3], [3 / 4, 1], [3 / 4, 2 / 3], [3 / 4, 1 / 3]])
    assert all(np.isclose(cs.length(i, "x"), i / 4) for i in range(4)) and all(np.isclose(cs.length(i, "y"), i / 3) for i in range(3))
    assert [int(cs.num_voxels(v, ax)) for v, ax in ((1, "x"), (1, "y"), (0.7, "x"), (0.6, "y"), (0.8, "x"), (0.7, "y"))] == [4, 3, 3, 2, 4, 3]
    for voxel in (np.array([1, 2]), [1, 2], da.Voxel([1, 2])):
        c = cs.coordinate(voxel)
        assert isinstance(c, da.Coordinate) and np.allclose(c, [0.5, 2 / 3])
    for voxels in (np.array([[1, 2], [2, 3]]), [[1, 2], [2, 3]], da.VoxelArray([[1, 2], [2, 3]])):
        c = cs.coordinate(voxels)
        assert isinstance(c, da.CoordinateArray) and np.allclose(c, [[0.5, 2 / 3], [0.75, 1 / 3]])
    for coordinate in (np.array([0.5, 2 / 3]), [0.5, 2 / 3], da.Coordinate([0.5, 2 / 3])):
        v = cs.voxel(coordinate)
        assert isinstance(v, da.Voxel) and np.allclose(v, [1, 2])
    pts = [[0.5, 2 / 3], [0.75, 1 / 3], [0.6, 0.9]]
    for coordinates in (np.array(pts), pts, da.CoordinateArray(pts)):
        v = cs.voxel(coordinates)
        assert isinstance(v, da.VoxelArray) and np.allclose(v, [[1, 2], [2, 3], [0, 2]])
    assert np.allclose(cs.coordinate_vector(np.array([1, 2])), [0.5, -1 / 3])
    assert np.allclose(cs.coordinate_vector(np.array([[1, 2], [2, 2]])), [[0.5, -1 / 3], [0.5, -2 / 3]])


def test_roi_masks_equal_the_reference_function():
    """roi_to_mask / zeros_like of the reference executed by tests/golden/make_golden_registration.py on a 37 x 53 x 3 image
    of 0.74 x 1.59 length units: metric and voxel ROIs (inside, partly outside, outside, the whole image, on voxel edges,
    reversed corners), both modes, a union and the empty list -- boolean masks, bit for bit."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "registration_golden.npz"))
    img = da.Image(np.zeros(tuple(g["roi/shape"])), dimensions=list(g["roi/dims"]))
    rois = {}
    for name in ("interior", "partly_outside", "outside", "full", "on_voxel_edges", "voxels", "voxels_reversed"):
        raw = g[f"roi/{name}_in"]
        rois[name] = da.VoxelArray(raw.astype(int)) if name.startswith("voxels") else da.CoordinateArray(raw)
        full = da.roi_to_mask(rois[name], img)
        vox = da.roi_to_mask(rois[name], img, mode="voxels")
        assert full.img.dtype == bool and np.array_equal(full.img, g[f"roi/{name}_shape_mask"])
        assert isinstance(vox, da.ScalarImage) and np.array_equal(vox.img, g[f"roi/{name}_voxels_mask"])
    union = da.roi_to_mask([rois["interior"], rois["voxels"], rois["partly_outside"]], img)
    assert np.array_equal(union.img, g["roi/union_mask"])
    assert np.array_equal(da.roi_to_mask([], img).img, g["roi/empty_list_mask"])
