"""oracle/curvature_numpy.map_coordinates against SciPy itself (the reference's third-party gather)."""
import numpy as np
import pytest
from scipy.ndimage import map_coordinates as sp_map

from oracle import curvature_numpy as cu


@pytest.mark.parametrize("dt", [np.float64, np.float32, np.uint8, np.uint16, np.int16])
@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("cdt", [np.float64, np.float32])
def test_map_coordinates_equals_scipy_bit_for_bit(dt, order, cdt):
    rng = np.random.default_rng(0)
    a = (rng.random((13, 17)) * (200 if np.dtype(dt).kind in "ui" else 1)).astype(dt)
    if np.dtype(dt).kind == "i":
        a = a - 100
    coords = np.stack([rng.uniform(-2, 15, 4000), rng.uniform(-2, 19, 4000)])
    special = np.array([[0, 0], [12, 16], [12.0, 3.5], [3.5, 16.0], [-0.0, 5], [12.0000001, 5], [-1e-9, 5], [11.999999, 16],
                        [0.5, 0.5], [6, 8], [np.nan, 3]]).T
    coords = np.concatenate([coords, special], axis=1).astype(cdt)
    np.testing.assert_array_equal(cu.map_coordinates(a, coords, order), sp_map(a, coords, order=order))


def test_single_row_and_column_images():
    rng = np.random.default_rng(1)
    for shape in ((1, 9), (7, 1)):
        a = rng.random(shape)
        coords = np.stack([rng.uniform(-0.5, shape[0], 300), rng.uniform(-0.5, shape[1], 300)])
        np.testing.assert_array_equal(cu.map_coordinates(a, coords, 1), sp_map(a, coords, order=1))


def test_transform_image_and_simple_correction_match_scipy_flow():
    rng = np.random.default_rng(2)
    img = rng.random((40, 56, 3)).astype(np.float32)
    kw = dict(horizontal_bulge=3e-6, vertical_bulge=-2e-6, horizontal_stretch=1e-6, vertical_stretch=-1.5e-6,
              horizontal_center_offset=3, vertical_center_offset=-2)
    out = cu.simple_curvature_correction(img, **kw)
    # the same flow with SciPy's gather
    X, Y = np.meshgrid(np.arange(56, dtype=np.float32), np.arange(40, dtype=np.float32))
    X, Y = cu.transform_coordinates(X, Y, **kw)
    grid = np.array([Y.ravel(), X.ravel()])
    assert grid.dtype == np.float32
    ref = np.stack([sp_map(img[..., c], grid, order=1).reshape(40, 56) for c in range(3)], axis=-1)
    np.testing.assert_array_equal(out, ref)
    assert not np.array_equal(out, img)
