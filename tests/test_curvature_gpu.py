"""Device gather (dxh_map_coordinates) against SciPy, and CurvatureCorrection against the NumPy / SciPy flow."""
import numpy as np
import pytest
from scipy.ndimage import map_coordinates as sp_map

import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
from oracle import curvature_numpy as cu
from oracle import warp_numpy as wo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [np.float64, np.float32, np.uint8, np.uint16])
@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("cdt", [np.float64, np.float32])
def test_map_coordinates_equals_scipy_bit_for_bit(dt, order, cdt):
    rng = np.random.default_rng(0)
    a = (rng.random((13, 17, 3)) * (250 if np.dtype(dt).kind == "u" else 1)).astype(dt)
    coords = np.stack([rng.uniform(-2, 15, 5000), rng.uniform(-2, 19, 5000)])
    special = np.array([[0, 0], [12, 16], [12.0, 3.5], [3.5, 16.0], [-0.0, 5], [12.0000001, 5], [-1e-9, 5], [11.999999, 16],
                        [0.5, 0.5], [6, 8], [np.nan, 3]]).T
    coords = np.concatenate([coords, special], axis=1).astype(cdt)
    out = da.map_coordinates(a, coords, order=order)
    assert out.dtype == a.dtype and out.shape == (coords.shape[1], 3)
    for c in range(3):
        np.testing.assert_array_equal(out[:, c], sp_map(a[..., c], coords, order=order))
    np.testing.assert_array_equal(da.map_coordinates(a[..., 0].copy(), coords, order=order), sp_map(a[..., 0], coords, order=order))


def test_map_coordinates_errors():
    a = np.zeros((4, 4), np.float32)
    with pytest.raises(NotImplementedError):
        da.map_coordinates(a, np.zeros((2, 3)), order=3)
    with pytest.raises(TypeError):
        da.map_coordinates(a.astype(np.int32), np.zeros((2, 3)))
    with pytest.raises(ValueError):
        da.map_coordinates(a, np.zeros((3, 3)))


KW = dict(horizontal_bulge=3e-7, vertical_bulge=-2e-7, horizontal_stretch=1e-7, vertical_stretch=-1.5e-7,
          horizontal_center_offset=5, vertical_center_offset=-3)


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
def test_simple_curvature_correction_matches_oracle(dtype):
    img = synthetic_image(300, 420, 3)
    if dtype == np.uint8:
        img = (img * 255).astype(np.uint8)
    cc = da.CurvatureCorrection(config={})
    out = cc.simple_curvature_correction(img, **KW)
    ref = cu.simple_curvature_correction(img, **KW)
    assert out.dtype == img.dtype and out.shape == img.shape
    np.testing.assert_array_equal(out, ref)
    assert not np.array_equal(out, img)


def test_correct_array_init_bulge_stretch_matches_oracle_and_caches_grid():
    img = synthetic_image(240, 320, 3)
    config = {"init": dict(horizontal_bulge=4e-7, vertical_bulge=1e-7),
              "bulge": dict(horizontal_bulge=-2e-7, vertical_bulge=3e-7, horizontal_center_offset=4),
              "stretch": dict(horizontal_stretch=2e-7, vertical_stretch=-1e-7, vertical_center_offset=2)}
    cc = da.CurvatureCorrection(config=config)
    out = cc(img)
    grid, shape = cu.precompute_grid(img.shape, config)
    np.testing.assert_array_equal(cc.cache["grid"], grid)
    assert tuple(cc.cache["shape"]) == tuple(shape)
    np.testing.assert_array_equal(out, cu.transform_image(img, grid, shape))
    # second image reuses the cached grid; scalar image in, scalar image out
    gray = np.ascontiguousarray(img[..., 0])
    np.testing.assert_array_equal(cc(gray), cu.transform_image(gray, grid, shape))


def test_crop_step_and_metadata():
    img = synthetic_image(200, 300, 3)
    crop = dict(pts_src=[[10, 8], [6, 190], [290, 195], [285, 12]], width=1.5, height=1.0)
    config = {"crop": crop, "bulge": dict(horizontal_bulge=2e-7)}
    cc = da.CurvatureCorrection(config=config)
    out = cc(da.OpticalImage(img, dimensions=[1.0, 1.5]))
    assert isinstance(out, da.OpticalImage) and out.img.shape == (200, 300, 3)
    assert out.dimensions == [1.0, 1.5] and list(np.asarray(out.origin)) == [0, 1.0]

    def crop_oracle(pixels, **kw):  # extract_quadrilateral_ROI with the restated OpenCV warp
        h, w = pixels.shape[:2]
        ar = kw["width"] / kw["height"]
        width, height = min(w, int(ar * float(h))), min(h, int(1.0 / ar * float(w)))
        dst = np.array([[0, 0], [0, height - 1], [width - 1, height - 1], [width - 1, 0]], np.float32)
        p = wo.get_perspective_transform(np.fliplr(np.asarray(kw["pts_src"], np.float32)), dst)  # dictionary configs: (row, col)
        return wo.warp_perspective(pixels.astype(np.float32), p, width, height).astype(pixels.dtype)

    grid, shape = cu.precompute_grid(img.shape, config, crop=crop_oracle)
    np.testing.assert_array_equal(cc.cache["grid"], grid)
    np.testing.assert_array_equal(out.img, cu.transform_image(img, grid, shape))


def test_full_size_gather_properties():
    """4000 x 8000 x 3 float32: the identity grid returns the image; a half-pixel shift is the neighbour mean."""
    img = synthetic_image(4000, 8000, 3)
    cc = da.CurvatureCorrection(config={})
    np.testing.assert_array_equal(cc.simple_curvature_correction(img), img)
    r, c = np.meshgrid(np.arange(100, dtype=np.float64) + 0.5, np.arange(200, dtype=np.float64), indexing="ij")
    out = da.map_coordinates(img, np.array([r.ravel(), c.ravel()])).reshape(100, 200, 3)
    ref = (img[:100, :200].astype(np.float64) * 0.5 + img[1:101, :200].astype(np.float64) * 0.5).astype(np.float32)
    np.testing.assert_array_equal(out, ref)


def test_device_resident_frames_and_grid():
    """A DeviceArray frame goes through the correction without leaving the device (same values as the host call); the
    object's cached coordinate grid is uploaded once and re-used; map_coordinates takes device grids."""
    img = synthetic_image(96, 128, 3)
    config = {"bulge": dict(horizontal_bulge=5e-6, vertical_bulge=-3e-6), "stretch": dict(horizontal_stretch=2e-6)}
    cc = da.CurvatureCorrection(config=config)
    want = cc(img)
    held = cc._grid_dev
    dev = cc(da.DeviceArray.from_host(img))
    assert isinstance(dev, da.DeviceArray) and dev.shape == want.shape and np.array_equal(dev.get(), want)
    assert cc._grid_dev is held                                   # no second upload of the grid
    gray = np.ascontiguousarray(img[..., 0])
    assert np.array_equal(cc(da.DeviceArray.from_host(gray)).get(), cc(gray))
    # chained: correction -> TV denoising, nothing in between on the host
    out = da.split_bregman_tvd(cc(da.DeviceArray.from_host(img)), mu=0.1, max_num_iter=5)
    assert np.array_equal(out.get(), da.split_bregman_tvd(want, mu=0.1, max_num_iter=5))
    # explicit device grids
    rng = np.random.default_rng(3)
    grid = np.stack([rng.uniform(-1, 97, 777), rng.uniform(-1, 129, 777)]).astype(np.float32)
    rows, cols = da.DeviceArray.from_host(grid[0]), da.DeviceArray.from_host(grid[1])
    assert np.array_equal(da.map_coordinates(img, (rows, cols)), da.map_coordinates(img, grid))
    assert np.array_equal(da.map_coordinates(da.DeviceArray.from_host(img), (rows, cols)).get(), da.map_coordinates(img, grid))


def _curv_golden():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "curvature_golden.npz"))


@pytest.mark.parametrize("tag", sorted({k.split("/")[0] for k in _curv_golden().files if k.endswith("/spec")}))
def test_product_against_outputs_of_the_reference_class(tag):
    """CurvatureCorrection on the device against what the reference's own class returned for the same image and
    configuration (tests/golden/make_golden_curvature.py): cached grid and corrected image, bit for bit."""
    import ast
    import copy

    from test_curvature_golden_cpu import CONFIGS

    g = _curv_golden()
    spec = ast.literal_eval(str(g[f"{tag}/spec"]))
    img = g[f"{tag}/img"]
    cc = da.CurvatureCorrection(config=copy.deepcopy(CONFIGS[spec["config"]]), interpolation_order=spec["order"],
                                resize_factor=spec["resize_factor"])
    out = cc(img)
    assert np.array_equal(cc.cache["grid"], g[f"{tag}/grid"]) and tuple(cc.cache["shape"]) == tuple(g[f"{tag}/shape"])
    want = g[f"{tag}/out"]
    assert out.dtype == want.dtype and np.array_equal(out, want)


def test_image_branch_against_the_reference_class():
    g = _curv_golden()
    cc = da.CurvatureCorrection(config={"bulge": dict(horizontal_bulge=5e-6, vertical_bulge=-3e-6)})
    im = da.ScalarImage(g["image/img"], name="frame")
    res = cc(im)
    assert bool(g["image/is_new_object"]) and res is not im and res.name == "frame"
    assert np.array_equal(res.img, g["image/out"])


def _roi_golden():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "warp_assembly_golden.npz"))


def test_extract_quadrilateral_roi_against_the_reference_function():
    """image/subregions.py:31-176 executed as it stands, its two OpenCV calls answered by the oracle's restatement
    (tests/golden/make_golden_warp_assembly.py): corner conventions (lists in both indexings, VoxelArray), target size from a
    shape or from physical width / height, destination points, and the cast back to the input's dtype -- bit for bit."""
    import warnings

    g = _roi_golden()
    src32, rgb8, rgb64 = g["roi/src32"], g["roi/src_rgb8"], g["roi/src_rgb64"]
    quad = g["roi/quad"].tolist()
    cases = {
        "default": lambda: da.extract_quadrilateral_ROI(src32),
        "list_reverse": lambda: da.extract_quadrilateral_ROI(src32, pts_src=quad),
        "list_matrix": lambda: da.extract_quadrilateral_ROI(src32, pts_src=[[q[1], q[0]] for q in quad], indexing="matrix"),
        "voxels": lambda: da.extract_quadrilateral_ROI(src32, pts_src=da.VoxelArray([[int(q[1]), int(q[0])] for q in quad])),
        "width_height": lambda: da.extract_quadrilateral_ROI(rgb8, pts_src=[[4, 4], [4, 58], [74, 60], [76, 2]], width=1.5, height=1.0),
        "shape_kw": lambda: da.extract_quadrilateral_ROI(rgb64, pts_src=[[3, 2], [2, 45], [60, 47], [62, 4]], shape=(40, 50)),
        "pts_dst": lambda: da.extract_quadrilateral_ROI(src32, pts_src=quad, pts_dst=[[0, 0], [0, 50], [70, 50], [70, 0]], shape=(55, 75)),
    }
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, call in cases.items():
            got, want = call(), g[f"roi/{name}"]
            assert got.shape == want.shape and got.dtype == want.dtype, name
            assert np.array_equal(got, want), name
        # "inter_area" has always run as INTER_LINEAR in the reference (subregions.py:157-158 never assigns the flag)
        assert np.array_equal(da.extract_quadrilateral_ROI(src32, pts_src=quad, interpolation="inter_area"), g["roi/list_reverse"])
        with pytest.raises(NotImplementedError):
            da.extract_quadrilateral_ROI(src32, pts_src=quad, interpolation="bicubic")
        # an image that is in HBM already stays there (target shape = source shape, the default)
        for name, kw in (("default", {}), ("list_reverse", dict(pts_src=quad))):
            d = da.extract_quadrilateral_ROI(da.DeviceArray.from_host(src32), **kw)
            assert isinstance(d, da.DeviceArray) and np.array_equal(d.get(), g[f"roi/{name}"]), name
        with pytest.raises(NotImplementedError):
            da.extract_quadrilateral_ROI(da.DeviceArray.from_host(src32), pts_src=quad, shape=(55, 75))


def test_extract_quadrilateral_roi_target_larger_than_the_source():
    img = synthetic_image(40, 60, 3)
    quad = [[5.0, 4.0], [4.0, 35.0], [55.0, 36.0], [56.0, 3.0]]
    got = da.extract_quadrilateral_ROI(img, pts_src=quad, shape=(90, 130))
    dst = np.array([[0, 0], [0, 89], [129, 89], [129, 0]], np.float32)
    want = wo.warp_perspective(img, wo.get_perspective_transform(np.asarray(quad, np.float32), dst), 130, 90)
    assert got.shape == (90, 130, 3) and np.array_equal(got, want)


def test_crop_step_against_the_reference_class():
    """CurvatureCorrection with a "crop" entry: the reference's class executed (its extract_quadrilateral_ROI calls answered as
    above); cached grid, corrected frame and corrected metadata."""
    import warnings

    g = _roi_golden()
    cfg = {"init": dict(horizontal_bulge=4e-7, vertical_bulge=1e-7),
           "crop": dict(pts_src=[[5, 4], [4, 62], [85, 64], [86, 3]], width=1.4, height=1.0),
           "bulge": dict(horizontal_bulge=-2e-7, vertical_bulge=3e-7)}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cc = da.CurvatureCorrection(config=cfg)
        out = cc(g["crop/frame"])
    assert np.array_equal(cc.cache["grid"], g["crop/grid"]) and tuple(cc.cache["shape"]) == tuple(g["crop/shape"])
    assert out.dtype == g["crop/out"].dtype and np.array_equal(out, g["crop/out"])
    meta = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in cc.correct_metadata().items()}
    assert repr(meta) == str(g["crop/metadata"])
