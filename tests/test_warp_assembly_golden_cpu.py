"""oracle/warp_numpy.py's assembly of the piecewise perspective warp against the reference's own find_and_warp loop
(tests/golden/make_golden_warp_assembly.py: the reference's PiecewisePerspectiveTransform, Patches and Image executed, with the
two OpenCV calls inside the loop answered by this repository's restatement -- see that script's header for what this does
and does not pin)."""
import os

import numpy as np
import pytest

import darsia_amd as da
from oracle import warp_numpy as W

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "warp_assembly_golden.npz"))
TAGS = sorted({k.split("/")[0] for k in G.files if k.endswith("/amp")})


def displacement_for(tag):
    img = G[f"{tag}/img"]
    h, w = img.shape[:2]
    amp = G[f"{tag}/amp"]

    def displacement(p):
        p = np.asarray(p, dtype=float)
        return np.array([amp[0] * np.sin(np.pi * p[:, 0] / w) * np.cos(2.0 * p[:, 1] / h), amp[1] * np.sin(np.pi * p[:, 1] / h)])

    return displacement


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_assembly_equals_the_reference_loop(tag):
    img = G[f"{tag}/img"]
    n = [int(v) for v in G[f"{tag}/n"]]
    patches = da.Patches(da.Image(img), n, rel_overlap=0.0)
    got = W.find_and_warp(img, patches.rois, patches.local_corners_voxels, patches.global_corners_voxels,
                          displacement_for(tag), bool(G[f"{tag}/reverse"]))
    want = G[f"{tag}/out"]
    assert got.dtype == want.dtype and np.array_equal(got, want)


def test_oracle_crop_step_equals_the_reference_class():
    """oracle/curvature_numpy.py's grid set-up with a "crop" entry against the grid the reference's CurvatureCorrection cached
    (its extract_quadrilateral_ROI executed as it stands, OpenCV calls answered by oracle/warp_numpy.py)."""
    from oracle import curvature_numpy as cu

    crop = dict(pts_src=[[5, 4], [4, 62], [85, 64], [86, 3]], width=1.4, height=1.0)
    cfg = {"init": dict(horizontal_bulge=4e-7, vertical_bulge=1e-7), "crop": crop,
           "bulge": dict(horizontal_bulge=-2e-7, vertical_bulge=3e-7)}

    def crop_oracle(pixels, **kw):
        h, w = pixels.shape[:2]
        ar = kw["width"] / kw["height"]
        width, height = min(w, int(ar * float(h))), min(h, int(1.0 / ar * float(w)))
        dst = np.array([[0, 0], [0, height - 1], [width - 1, height - 1], [width - 1, 0]], np.float32)
        p = W.get_perspective_transform(np.fliplr(np.asarray(kw["pts_src"], np.float32)), dst)  # dictionary configs: (row, col)
        return W.warp_perspective(pixels.astype(np.float32), p, width, height).astype(pixels.dtype)

    frame = G["crop/frame"]
    grid, shape = cu.precompute_grid(frame.shape, cfg, crop=crop_oracle)
    assert np.array_equal(grid, G["crop/grid"]) and tuple(shape) == tuple(G["crop/shape"])
    assert np.array_equal(cu.transform_image(frame, grid, shape), G["crop/out"])


def test_dictionary_configuration_is_normalised_as_in_the_reference():
    """curvature.py:97-164 and 827-855 executed by the generator: the keys kept and dropped, the defaults, the crop corners as a
    VoxelArray -- and as a plain list again once resize_factor has scaled them."""
    import warnings

    messy = {"init": dict(horizontal_bulge=1e-7, horizontal_center_offset=9), "crop": dict(pts_src=[[1, 2], [3, 40], [50, 41], [52, 3]]),
             "stretch": dict(vertical_stretch=2e-7), "use_cache": True, "cache": "somewhere.npy"}

    def plain(cfg):
        return {k: {kk: (np.asarray(vv).tolist() if kk == "pts_src" else vv) for kk, vv in v.items()} for k, v in cfg.items()}

    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        norm = da.CurvatureCorrection(config=messy)
        scaled = da.CurvatureCorrection(config=messy, resize_factor=0.5)
    assert any("curvature.bulge" in str(w.message) for w in caught)
    assert repr(plain(norm.config)) == str(G["cfg/normalised"]) and type(norm.config["crop"]["pts_src"]).__name__ == str(G["cfg/pts_type"])
    assert norm.use_cache == bool(G["cfg/use_cache"])
    assert repr(plain(scaled.config)) == str(G["cfg/scaled"])
    assert type(scaled.config["crop"]["pts_src"]).__name__ == str(G["cfg/scaled_pts_type"])
