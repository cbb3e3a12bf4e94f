"""Golden vectors for the ASSEMBLY of the piecewise perspective warp (corrections/shape/piecewiseperspective.py:35-145):
the reference's own ``PiecewisePerspectiveTransform.find_and_warp`` and ``darsia.Patches`` are executed on the reference's own
``Image`` class -- patch loop, corner displacement, effective origin / size of every warped tile, clipping to the canvas,
additive seams, cast back to the input dtype, metadata of the result.

The two OpenCV functions that loop calls, ``cv2.getPerspectiveTransform`` and ``cv2.warpPerspective(..., INTER_LINEAR)``, do
not exist here.  They are provided by a ``cv2`` module backed by THIS repository's restatement of OpenCV's published
algorithms (oracle/warp_numpy.py: get_perspective_transform, warp_perspective).  These vectors therefore pin everything
around those two calls on the reference's code, and nothing about the two calls themselves, whose parity with OpenCV
stays unpinned (oracle/warp_numpy.py header).

The same for ``extract_quadrilateral_ROI`` (image/subregions.py:31-176: corner conventions, target size from a shape or from
physical width / height, destination points, cast back to the input dtype) and for the "crop" step of ``CurvatureCorrection``
(corrections/shape/curvature.py), whose coordinate images pass through it.

    /opt/conda/bin/python3.9 tests/golden/make_golden_warp_assembly.py      # -> warp_assembly_golden.npz

TEST INFRASTRUCTURE, build container only.  The output holds inputs and expected outputs (data, no source).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    from make_golden_image import _load, reference_image_classes
    from oracle import warp_numpy as W

    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1
    cv2.getPerspectiveTransform = lambda src, dst: W.get_perspective_transform(np.asarray(src), np.asarray(dst))

    def warp_perspective(src, m, dsize, flags=1):
        assert flags == cv2.INTER_LINEAR
        return W.warp_perspective(np.asarray(src, dtype=np.float32), np.asarray(m, dtype=np.float64), int(dsize[0]), int(dsize[1]))

    cv2.warpPerspective = warp_perspective
    sys.modules["cv2"] = cv2
    da = reference_image_classes()
    _load(da, "image/patches.py")
    _load(da, "corrections/shape/piecewiseperspective.py")

    rng = np.random.default_rng(23)
    out = {"meta/numpy": np.array(np.__version__)}
    cases = [("gray_f32", (80, 120), np.float32, [3, 4], False), ("rgb_f32_rev", (64, 96, 3), np.float32, [2, 3], True),
             ("gray_u8", (75, 101), np.uint8, [3, 5], False), ("rgb_f64", (60, 84, 3), np.float64, [4, 4], False)]
    for tag, shape, dtype, n, reverse in cases:
        raw = rng.random(shape)
        img = (raw * 255).astype(dtype) if np.dtype(dtype).kind == "u" else raw.astype(dtype)
        h, w = shape[:2]
        amp = np.array([5.0, 3.5])

        def displacement(p, h=h, w=w, amp=amp):
            p = np.asarray(p, dtype=float)
            return np.array([amp[0] * np.sin(np.pi * p[:, 0] / w) * np.cos(2.0 * p[:, 1] / h), amp[1] * np.sin(np.pi * p[:, 1] / h)])

        image_cls = da.ScalarImage if img.ndim == 2 else da.OpticalImage
        base = image_cls(img.copy(), dimensions=[0.5 * h / 100, 0.5 * w / 100], name=tag)
        patches = da.Patches(base, n, rel_overlap=0.0)
        res = da.PiecewisePerspectiveTransform().find_and_warp(patches, displacement, reverse)
        out[f"{tag}/img"], out[f"{tag}/out"] = img, res.img
        out[f"{tag}/n"], out[f"{tag}/reverse"], out[f"{tag}/amp"] = np.array(n), np.array(reverse), amp
        out[f"{tag}/meta"] = np.array([type(res).__name__, str(res.name), str(list(res.dimensions))])
    # ---- extract_quadrilateral_ROI (image/subregions.py:31-176) and the crop step of CurvatureCorrection
    cv2.INTER_NEAREST, cv2.INTER_AREA = 0, 3
    if "tomllib" not in sys.modules:
        try:
            import tomllib  # noqa: F401
        except Exception:  # noqa: BLE001
            sys.modules["tomllib"] = types.ModuleType("tomllib")
    _load(da, "image/subregions.py")
    _load(da, "corrections/basecorrection.py")
    _load(da, "corrections/shape/curvature.py")
    import warnings
    warnings.simplefilter("ignore")
    src32 = rng.random((70, 90)).astype(np.float32)
    src_rgb8 = (rng.random((64, 80, 3)) * 255).astype(np.uint8)
    src_rgb64 = rng.random((50, 66, 3))
    out["roi/src32"], out["roi/src_rgb8"], out["roi/src_rgb64"] = src32, src_rgb8, src_rgb64
    quad = [[8.0, 5.0], [6.0, 60.0], [80.0, 64.0], [84.0, 3.0]]            # (col, row): reverse matrix indexing
    out["roi/quad"] = np.array(quad)
    out["roi/default"] = da.extract_quadrilateral_ROI(src32)
    out["roi/list_reverse"] = da.extract_quadrilateral_ROI(src32, pts_src=quad)
    out["roi/list_matrix"] = da.extract_quadrilateral_ROI(src32, pts_src=[[q[1], q[0]] for q in quad], indexing="matrix")
    out["roi/voxels"] = da.extract_quadrilateral_ROI(src32, pts_src=da.VoxelArray([[int(q[1]), int(q[0])] for q in quad]))
    out["roi/width_height"] = da.extract_quadrilateral_ROI(src_rgb8, pts_src=[[4, 4], [4, 58], [74, 60], [76, 2]], width=1.5, height=1.0)
    out["roi/shape_kw"] = da.extract_quadrilateral_ROI(src_rgb64, pts_src=[[3, 2], [2, 45], [60, 47], [62, 4]], shape=(40, 50))
    out["roi/pts_dst"] = da.extract_quadrilateral_ROI(src32, pts_src=quad, pts_dst=[[0, 0], [0, 50], [70, 50], [70, 0]], shape=(55, 75))
    crop_cfg = {"init": dict(horizontal_bulge=4e-7, vertical_bulge=1e-7),
                "crop": dict(pts_src=[[5, 4], [4, 62], [85, 64], [86, 3]], width=1.4, height=1.0),
                "bulge": dict(horizontal_bulge=-2e-7, vertical_bulge=3e-7)}
    cc = da.CurvatureCorrection(config=crop_cfg)
    frame = rng.random((70, 90, 3)).astype(np.float32)
    res = cc(frame)
    out["crop/frame"], out["crop/out"] = frame, res
    out["crop/grid"], out["crop/shape"] = np.array(cc.cache["grid"]), np.array(cc.cache["shape"])
    out["crop/metadata"] = np.array(repr({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in cc.correct_metadata().items()}))
    # what the reference makes of a dictionary configuration (curvature.py:97-164), incl. keys it drops and defaults it adds
    messy = {"init": dict(horizontal_bulge=1e-7, horizontal_center_offset=9), "crop": dict(pts_src=[[1, 2], [3, 40], [50, 41], [52, 3]]),
             "stretch": dict(vertical_stretch=2e-7), "use_cache": True, "cache": "somewhere.npy"}
    plain = lambda cfg: {k: {kk: (np.asarray(vv).tolist() if kk == "pts_src" else vv) for kk, vv in v.items()} for k, v in cfg.items()}
    norm = da.CurvatureCorrection(config=messy)
    out["cfg/normalised"] = np.array(repr(plain(norm.config)))
    out["cfg/pts_type"] = np.array(type(norm.config["crop"]["pts_src"]).__name__)
    out["cfg/use_cache"] = np.array(norm.use_cache)
    scaled = da.CurvatureCorrection(config=messy, resize_factor=0.5)
    out["cfg/scaled"] = np.array(repr(plain(scaled.config)))
    out["cfg/scaled_pts_type"] = np.array(type(scaled.config["crop"]["pts_src"]).__name__)
    np.savez_compressed(os.path.join(HERE, "warp_assembly_golden.npz"), **out)
    print("wrote", len(out), "arrays;", {k: v.dtype.name for k, v in out.items() if k.endswith("/out")})


if __name__ == "__main__":
    main()
