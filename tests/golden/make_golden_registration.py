"""Golden vectors for the host-side geometry of the registration path, produced by running the reference's own code:

    utils/point.py, image/indexing.py, image/coordinatesystem.py    Voxel / Coordinate types, CoordinateSystem
    image/patches.py                                                 Patches (ROIs, centres, corners)          (:28-304)
    multi_image_analysis/translationanalysis.py                      TranslationAnalysis: spline data points, boundary
                                                                     points, return_patch_translation, add_ / deduct_
    multi_image_analysis/imageregistration.py                        DiffeomorphicImageRegistration.evaluate / displacement
    utils/standard_images.py                                         roi_to_mask, zeros_like                   (:12-137)

loaded file by file onto an empty ``darsia`` shim (make_golden.py's way), each source executed unmodified with the
``annotations`` future flag.  The images are the reference's own ``darsia.Image`` (image/image.py, loaded as make_golden_image.py does: empty ``cv2`` and
``plotly`` modules satisfy its plotting / I-O imports).  What stands in for the parts of DarSIA that need OpenCV:
  * the per-patch translation estimator (ORB + RANSAC in OpenCV, SURVEY.md section 2: out of scope): a table that hands out
    prescribed translations in the order the analysis visits the patches;
  * ``cv2`` itself is not needed by any of these files.
Run in the build container under /opt/conda's Python 3.9 (patches.py imports scikit-image; SciPy 1.7.1 has
RBFInterpolator):

    /opt/conda/bin/python3.9 tests/golden/make_golden_registration.py     # -> registration_golden.npz

TEST INFRASTRUCTURE.  The output holds inputs and expected outputs (data, no source).
"""
from __future__ import annotations

import copy
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


def main():
    sys.path.insert(0, HERE)
    from make_golden_image import reference_image_classes

    # the reference's own Image / ScalarImage / CoordinateSystem / point types (image/image.py executed with empty ``cv2`` and
    # ``plotly`` modules for its plotting / I-O imports; see make_golden_image.py)
    da = reference_image_classes()
    Image = da.Image
    _load(da, "image/patches.py")
    _load(da, "multi_image_analysis/translationanalysis.py")

    class Table:
        def __init__(self, table):
            self.table, self.k = list(table), 0

        def find_effective_translation(self, img_src, img_dst, roi_src, roi_dst, mask_src=None, mask_dst=None, plot_matches=False):
            t = self.table[self.k % len(self.table)]
            self.k += 1
            if t is None:
                return None, False
            return np.array([[1.0, 0.0, t[0]], [0.0, 1.0, t[1]]]), True

    out = {"meta/numpy": np.array(np.__version__)}
    rng = np.random.default_rng(5)
    # ---- Patches geometry
    for tag, shape, dims, n, ov in (("p0", (100, 130), [1.0, 1.3], [3, 4], 0.1), ("p1", (120, 160), [1.2, 1.6], [3, 4], 0.0),
                                    ("p2", (97, 211), [0.5, 1.1], [5, 7], 0.25), ("p3", (64, 64), [2.0, 1.0], [1, 2], 0.05)):
        im = Image(rng.random(shape), dimensions=dims)
        p = da.Patches(im, n, rel_overlap=ov)
        out[f"{tag}/shape"], out[f"{tag}/dims"], out[f"{tag}/n"], out[f"{tag}/ov"] = np.array(shape), np.array(dims), np.array(n), np.array(ov)
        out[f"{tag}/rois"] = np.array([[[r.start, min(r.stop, shape[0]), c.start, min(c.stop, shape[1])] for (r, c) in row] for row in p.rois])
        out[f"{tag}/centers_cartesian"] = np.array(p.global_centers_cartesian, dtype=float)
        out[f"{tag}/centers_voxels"] = np.array(p.global_centers_voxels)
        out[f"{tag}/corners_voxels"] = np.array(p.global_corners_voxels)
        out[f"{tag}/local_corners_voxels"] = np.array(p.local_corners_voxels)
        out[f"{tag}/corners_cartesian"] = np.array(p.global_corners_cartesian, dtype=float)
        out[f"{tag}/patch_1_shape"] = np.array(p(n[0] - 1, n[1] - 1).img.shape)
    # ---- TranslationAnalysis
    base = Image(rng.random((100, 130)).astype(np.float32), dimensions=[1.0, 1.3])
    n = [3, 4]
    table = [(1.0 + 0.1 * k, -2.0 + 0.3 * k) for k in range(12)]
    table[5] = None
    ta = da.TranslationAnalysis(base, N_patches=n, rel_overlap=0.1, translation_estimator=Table(table))
    ta.load_image(base)
    translation, ok = ta.find_translation()
    out["ta/table"] = np.array([(np.nan, np.nan) if t is None else t for t in table])
    out["ta/ok"], out["ta/have"] = np.array(ok), ta.have_translation
    out["ta/x_points"], out["ta/x_values"] = np.array(ta.interpolator_translation_x.y), np.array(ta.interpolator_translation_x.d).reshape(-1)
    out["ta/y_points"], out["ta/y_values"] = np.array(ta.interpolator_translation_y.y), np.array(ta.interpolator_translation_y.d).reshape(-1)
    bx, bxv = ta.bc_x(["pixel", "pixel"])
    by, byv = ta.bc_y(["pixel", "pixel"])
    out["ta/bc_x"], out["ta/bc_y"] = np.array(bx, dtype=float), np.array(by, dtype=float)
    q = rng.random((9, 2)) * np.array([130, 100])
    out["ta/query"], out["ta/query_out"] = q, np.array(translation(q))
    for rev in (True, False):
        for units in ("metric", "pixel"):
            out[f"ta/patch_translation_rev{int(rev)}_{units}"] = np.array(ta.return_patch_translation(rev, units))
    # composition: does the reference's code run (it appends (1,)-arrays and floats to one list)?
    other = da.TranslationAnalysis(base, N_patches=[2, 3], rel_overlap=0.1,
                                   translation_estimator=Table([(1.5, 0.7), (-0.5, 0.2), (0.3, -0.9), (0.8, 0.1), (-1.1, 0.4), (0.6, -0.3)]))
    other.load_image(base)
    other.find_translation()
    out["comp/other_x_points"], out["comp/other_x_values"] = np.array(other.interpolator_translation_x.y), np.array(other.interpolator_translation_x.d).reshape(-1)
    out["comp/other_y_points"], out["comp/other_y_values"] = np.array(other.interpolator_translation_y.y), np.array(other.interpolator_translation_y.d).reshape(-1)
    for op in ("add", "deduct"):
        t2 = da.TranslationAnalysis(base, N_patches=n, rel_overlap=0.1, translation_estimator=Table(table))
        t2.load_image(base)
        t2.find_translation()
        try:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                getattr(t2, f"{op}_translation_analysis")(other)
            out[f"comp/{op}_status"] = np.array("ok")
            out[f"comp/{op}_x_points"] = np.array(t2.interpolator_translation_x.y)
            out[f"comp/{op}_x_values"] = np.array(t2.interpolator_translation_x.d, dtype=float).reshape(-1)
            out[f"comp/{op}_y_values"] = np.array(t2.interpolator_translation_y.d, dtype=float).reshape(-1)
            out[f"comp/{op}_query_out"] = np.array(t2.translation(q), dtype=float)
        except Exception as e:  # noqa: BLE001 -- recorded: what the reference does under this NumPy
            out[f"comp/{op}_status"] = np.array(f"{type(e).__name__}: {e}"[:300])
    # ---- DiffeomorphicImageRegistration: evaluate / displacement (imageregistration.py:166-306)
    _load(da, "multi_image_analysis/imageregistration.py")
    reg_table = [(0.8 * np.sin(k), 1.1 * np.cos(0.7 * k)) for k in range(6)]
    da.TranslationEstimator = lambda max_features, tol: Table(reg_table)   # the ORB / RANSAC estimator's stand-in
    dst = Image(rng.random((40, 52)).astype(np.float32), dimensions=[0.8, 1.04])
    reg = da.DiffeomorphicImageRegistration(dst, N_patches=[2, 3], rel_overlap=0.1)
    reg.translation_analysis.load_image(dst)
    reg.translation_analysis.find_translation()
    out["reg/table"] = np.array(reg_table)
    pts_metric = np.c_[rng.random(11) * 1.04, rng.random(11) * 0.8]
    pts_pixel = np.c_[rng.random(11) * 52, rng.random(11) * 40]
    out["reg/pts_metric"], out["reg/pts_pixel"] = pts_metric, pts_pixel
    for rev in (False, True):
        out[f"reg/evaluate_metric_rev{int(rev)}"] = np.array(reg.evaluate(da.CoordinateArray(pts_metric), rev, "metric"), dtype=float)
        out[f"reg/evaluate_pixel_rev{int(rev)}"] = np.array(reg.evaluate(pts_pixel.copy(), rev, "pixel"), dtype=float)
    out["reg/evaluate_patches"] = np.array(reg.evaluate(da.Patches(dst, [2, 2])), dtype=float)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        out["reg/displacement"] = np.array(reg.displacement(), dtype=float)
    # ---- ImageRegistration facade: keyword handling (imageregistration.py:490-606); construction only
    facade = {}
    kw_sets = {
        "flat": dict(N_patches=[2, 3]),
        "nested": dict(N_patches=[[2, 3], [4, 6]], rel_overlap=0.1),
        "nested_lists": dict(N_patches=[[1, 2], [2, 3], [4, 4]], rel_overlap=[0.0, 0.1, 0.2], max_features=[100, 200, 300], tol=[0.1, 0.05, 0.02]),
        "onelevel_of_nested": dict(method="onelevel", N_patches=[[2, 3], [4, 6]], rel_overlap=[0.3, 0.1], tol=[0.2, 0.1], max_features=[50, 60]),
        "multilevel_of_flat_grid_list": dict(method="multilevel", N_patches=[[3, 3]], rel_overlap=0.25),
        "bad_int_overlap": dict(N_patches=[[2, 3], [4, 6]], rel_overlap=0),
        "bad_length": dict(N_patches=[[2, 3], [4, 6]], rel_overlap=[0.1]),
        "bad_tol_type": dict(N_patches=[[2, 3], [4, 6]], tol=1),
    }
    for name, kw in kw_sets.items():
        kw = dict(kw)
        method = kw.pop("method", None)
        try:
            r = da.ImageRegistration(dst, method, **kw)
            inner = r.image_registration
            if type(inner).__name__ == "MultiscaleDiffeomorphicImageRegistration":
                facade[name] = ("multilevel", [dict(c) for c in inner.config])
            else:
                facade[name] = ("onelevel", dict(N_patches=list(inner.N_patches), rel_overlap=inner.rel_overlap))
        except Exception as e:  # noqa: BLE001
            facade[name] = ("error", type(e).__name__ + ": " + str(e))
    out["facade/repr"] = np.array(repr(facade))
    # ---- roi_to_mask (utils/standard_images.py:73-137): integer / boolean work, bit-exact
    da.ROI = type("ROI", (), {})   # polygon ROIs live in image/roi.py (shapely); only named in an isinstance test here
    ref_img = Image(rng.random((37, 53, 3)), dimensions=[0.74, 1.59], scalar=False)
    rois = {
        "interior": da.CoordinateArray([[0.31, 0.52], [0.93, 0.18]]),
        "partly_outside": da.CoordinateArray([[-0.2, 0.9], [0.4, 0.5]]),
        "outside": da.CoordinateArray([[2.0, 2.0], [3.0, 3.0]]),
        "full": da.CoordinateArray([[0.0, 0.74], [1.59, 0.0]]),
        "on_voxel_edges": da.CoordinateArray([[0.30, 0.60], [0.60, 0.30]]),
        "voxels": da.VoxelArray([[5, 7], [20, 41]]),
        "voxels_reversed": da.VoxelArray([[30, 50], [10, 2]]),
    }
    out["roi/shape"], out["roi/dims"] = np.array(ref_img.img.shape), np.array(ref_img.dimensions)
    for name, roi in rois.items():
        out[f"roi/{name}_in"] = np.array(roi, dtype=float)
        out[f"roi/{name}_shape_mask"] = np.array(da.roi_to_mask(roi, ref_img).img)
        out[f"roi/{name}_voxels_mask"] = np.array(da.roi_to_mask(roi, ref_img, mode="voxels").img)
    out["roi/union_mask"] = np.array(da.roi_to_mask([rois["interior"], rois["voxels"], rois["partly_outside"]], ref_img).img)
    out["roi/empty_list_mask"] = np.array(da.roi_to_mask([], ref_img).img)
    np.savez_compressed(os.path.join(HERE, "registration_golden.npz"), **out)
    print("wrote", len(out), "arrays; add:", out["comp/add_status"], "deduct:", out["comp/deduct_status"])


if __name__ == "__main__":
    main()
