"""Device resize (dxh_resize through Resize / cv_resize) against the cv2.resize restatement: bit-exact."""
import numpy as np
import pytest

import darsia_amd as da
from darsia_amd import resizing as rz
from darsia_amd.synthetic import synthetic_image
from oracle import resize_numpy as R

pytestmark = pytest.mark.gpu

CASES = [dict(dsize=(37, 29)), dict(dsize=(120, 90)), dict(fx=0.5, fy=0.5), dict(fx=0.37, fy=0.61), dict(fx=2.0, fy=1.5),
         dict(dsize=(16, 12)), dict(fx=1.0, fy=1.0)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.uint8])
@pytest.mark.parametrize("interp", [R.INTER_NEAREST, R.INTER_LINEAR, R.INTER_AREA])
@pytest.mark.parametrize("kw", CASES)
def test_cv_resize_matches_restatement_bit_exact(dtype, interp, kw):
    img = synthetic_image(48, 64, 3)
    img = (img * 255).astype(np.uint8) if dtype == np.uint8 else img.astype(dtype)
    out = rz.cv_resize(img, interpolation=interp, **kw)
    ref = np.stack([R.resize(np.ascontiguousarray(img[..., c]), interpolation=interp, **kw) for c in range(3)], axis=-1)
    assert out.dtype == img.dtype and out.shape == ref.shape
    np.testing.assert_array_equal(out, ref)


def test_odd_size_integer_area_and_gray():
    img = np.ascontiguousarray(synthetic_image(47, 63, 3)[..., 0])
    out = rz.cv_resize(img, fx=0.5, fy=0.5, interpolation=R.INTER_AREA)
    np.testing.assert_array_equal(out, R.resize(img, fx=0.5, fy=0.5, interpolation=R.INTER_AREA))


def test_resize_object_keywords_conservative_and_images():
    img = synthetic_image(60, 80, 3)
    r = da.Resize(key="example ", **{"example resize x": 0.5, "example resize y": 0.25, "example resize interpolation": "inter_area",
                                      "example resize conservative": True})
    out = r(img)
    assert out.shape == (15, 40, 3) and (r.fx, r.fy, r.interpolation, r.is_conservative) == (0.5, 0.25, rz.INTER_AREA, True)
    np.testing.assert_allclose(out.sum(), img.sum(), rtol=1e-5)  # conservative: the sum survives
    im = da.OpticalImage(img, dimensions=[0.6, 0.8])
    small = da.resize(im, shape=(30, 40), interpolation="inter_area")
    assert isinstance(small, da.OpticalImage) and small.img.shape == (30, 40, 3) and small.dimensions == [0.6, 0.8]
    np.testing.assert_allclose(small.img, img.reshape(30, 2, 40, 2, 3).mean((1, 3)), rtol=1e-6)
    same = da.Resize(ref_image=im)(img)
    np.testing.assert_array_equal(same, img)
    with pytest.raises(NotImplementedError):
        da.Resize(interpolation="inter_cubic")
    eq = da.equalize_voxel_size(da.OpticalImage(img, dimensions=[0.6, 1.6]))  # voxels 0.01 x 0.02 -> 0.01
    assert eq.img.shape[:2] == (60, 160)


def test_save_load_roundtrip(tmp_path):
    r = da.Resize(shape=(30, 40), interpolation="inter_nearest", dtype=np.float32)
    r.save(tmp_path / "r.npz")
    q = da.Resize()
    q.load(tmp_path / "r.npz")
    assert (q.shape, q.dsize, q.interpolation) == ((30, 40), (40, 30), rz.INTER_NEAREST)


def test_uniform_refinement_host_utility():
    im = da.ScalarImage(np.arange(12, dtype=np.float64).reshape(3, 4), dimensions=[3.0, 4.0])
    fine = da.uniform_refinement(im, 1)
    np.testing.assert_array_equal(fine.img, np.repeat(np.repeat(im.img, 2, 0), 2, 1))
    coarse = da.uniform_refinement(da.ScalarImage(np.arange(16, dtype=np.float64).reshape(4, 4)), -1)
    np.testing.assert_array_equal(coarse.img, np.arange(16.).reshape(2, 2, 2, 2).mean((1, 3)))


def test_full_size_downsample_for_tvd():
    """8000 x 4000 RGB -> quarter size by INTER_AREA equals the 2 x 2 block mean."""
    img = synthetic_image(4000, 8000, 3)
    out = rz.cv_resize(img, fx=0.5, fy=0.5, interpolation=rz.INTER_AREA)
    ref = ((img[0::2, 0::2] + img[0::2, 1::2]) + img[1::2, 0::2] + img[1::2, 1::2]) * np.float32(0.25)
    np.testing.assert_array_equal(out, ref)


def test_reference_resize_unit_tests():
    """tests/unit/test_resize.py of the reference, re-expressed: equalize_voxel_size (default and explicit voxel size)
    keeps the dimensions and makes the voxels square; resizing an ExtensiveImage conserves its sum by default, through
    the function and through the object."""
    image = da.Image(np.ones((10, 10)), width=2, height=1, scalar=True)
    assert not np.isclose(image.voxel_size[0], image.voxel_size[1])
    for kwargs in ({}, {"voxel_size": 0.05}):
        resized = da.equalize_voxel_size(image, **kwargs)
        assert np.isclose(resized.voxel_size[0], resized.voxel_size[1])
        assert np.allclose(image.dimensions, resized.dimensions)
    ext = da.ExtensiveImage(np.arange(1, 17, dtype=float).reshape((4, 4)), dimensions=[1, 1], scalar=True)
    for out in (da.resize(ext, shape=(2, 2)), da.Resize(shape=(2, 2))(ext)):
        assert isinstance(out, da.ExtensiveImage) and out.img.shape == (2, 2)
        assert np.isclose(np.sum(out.img), np.sum(ext.img))
    # an explicit choice wins over the image type (restoration/resize.py:186-188)
    plain = da.Resize(shape=(2, 2), **{"resize conservative": False})(ext)
    assert np.isclose(np.sum(plain.img), np.sum(ext.img) / 4)


def test_resize_assembly_against_the_reference_class():
    """The reference's own Resize / resize / equalize_voxel_size executed on its own Image classes, with cv2.resize / split /
    merge answered by the oracle's restatement (tests/golden/make_golden_resize_assembly.py): keyword handling and attributes,
    channel handling, dtype conversion, conservative rescaling, Image re-wrapping -- results bit for bit."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resize_assembly_golden.npz"))
    ref = da.OpticalImage(np.zeros((30, 40, 3)), dimensions=[0.3, 0.4])
    cases = {
        "keyed": (dict(key="example ", **{"example resize x": 0.5, "example resize y": 0.25, "example resize interpolation": "inter_area",
                                           "example resize conservative": True}), "rgb"),
        "general_factor": (dict(**{"resize": 0.5, "resize interpolation": "inter_linear"}), "rgb"),
        "shape_nearest": (dict(shape=(33, 47), interpolation="inter_nearest"), "gray64"),
        "shape_default_interp": (dict(shape=(90, 120)), "rgb"),
        "fx_fy_area": (dict(fx=0.5, fy=0.5, interpolation="inter_area"), "u8"),
        "dtype_before": (dict(fx=0.5, fy=0.5, interpolation="inter_area", dtype=np.float32), "u8"),
        "ref_image": (dict(ref_image=ref, interpolation="inter_linear"), "rgb"),
        "identity": (dict(), "gray64"),
    }
    for name, (kw, which) in cases.items():
        r = da.Resize(**kw)
        attrs = repr((r.shape, r.dsize, r.fx, r.fy, r.interpolation, r.is_conservative, r._conservative_explicit))
        assert attrs == str(g[f"{name}/attrs"]), name
        out = r(g[f"in/{which}"])
        want = g[f"{name}/out"]
        assert out.dtype == want.dtype and out.shape == want.shape and np.array_equal(out, want), name
    assert str(g["bad_interpolation"]) == "NotImplementedError"
    im = da.OpticalImage(g["in/rgb"].astype(np.float64), dimensions=[0.6, 0.8], name="photo")
    small = da.resize(im, shape=(30, 40), interpolation="inter_area")
    assert [type(small).__name__, str(small.name), str(list(small.dimensions))] == list(g["image/meta"])
    assert np.array_equal(small.img, g["image/out"])
    ext = da.ExtensiveImage(np.arange(1.0, 17.0).reshape(4, 4), dimensions=[1, 1], scalar=True)
    assert np.array_equal(da.Resize(shape=(2, 2))(ext).img, g["ext/default"])
    assert np.array_equal(da.Resize(shape=(2, 2), **{"resize conservative": False})(ext).img, g["ext/explicit_off"])
    assert np.array_equal(da.resize(ext, shape=(2, 2)).img, g["ext/function"])
    eq = da.equalize_voxel_size(da.OpticalImage(g["in/rgb"].astype(np.float64), dimensions=[0.6, 1.6]))
    assert tuple(eq.img.shape) == tuple(g["eq/default_shape"]) and np.array_equal(eq.img, g["eq/default_out"])
    eq2 = da.equalize_voxel_size(da.Image(np.ones((10, 10)), width=2, height=1, scalar=True), voxel_size=0.05)
    assert tuple(eq2.img.shape) == tuple(g["eq/explicit_shape"])
