"""Golden vectors for the ASSEMBLY around ``cv2.resize`` in the reference's ``Resize`` / ``resize`` / ``equalize_voxel_size``
(restoration/resize.py:18-320): keyword handling (``key`` prefixes, "resize" / "resize x" / "resize y" / "resize shape" /
"resize interpolation" / "resize dtype" / "resize conservative"), reference image, channel splitting of arrays with more than
two axes, dtype conversion before resizing, conservative rescaling (explicit and by image type), re-wrapping of Images with
their metadata.  The reference's own classes are executed on its own Image classes.

``cv2.resize`` / ``cv2.split`` / ``cv2.merge`` do not exist here: a ``cv2`` module backed by THIS repository's restatement of
OpenCV's resize (oracle/resize_numpy.py) answers them.  These vectors pin what surrounds the call, not ``cv2.resize`` itself,
whose parity with OpenCV stays unpinned (oracle/resize_numpy.py header).

    /opt/conda/bin/python3.9 tests/golden/make_golden_resize_assembly.py     # -> resize_assembly_golden.npz

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
    from oracle import resize_numpy as R

    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.INTER_AREA = R.INTER_NEAREST, R.INTER_LINEAR, R.INTER_AREA
    cv2.split = lambda a: tuple(np.ascontiguousarray(a[..., c]) for c in range(a.shape[2])) if a.ndim == 3 else (a,)
    cv2.merge = lambda chans: chans[0] if len(chans) == 1 else np.stack(chans, axis=-1)
    cv2.resize = lambda src, dsize=None, fx=None, fy=None, interpolation=R.INTER_LINEAR: R.resize(
        np.asarray(src), dsize=dsize, fx=fx, fy=fy, interpolation=interpolation)
    sys.modules["cv2"] = cv2
    da = reference_image_classes()
    _load(da, "restoration/resize.py")

    rng = np.random.default_rng(41)
    out = {"meta/numpy": np.array(np.__version__)}
    rgb = rng.random((60, 80, 3)).astype(np.float32)
    gray64 = rng.random((45, 70))
    u8 = (rng.random((40, 56, 3)) * 255).astype(np.uint8)
    out["in/rgb"], out["in/gray64"], out["in/u8"] = rgb, gray64, u8
    ref = da.OpticalImage(rng.random((30, 40, 3)), dimensions=[0.3, 0.4])
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
        src = out[f"in/{which}"]
        res = r(src)
        out[f"{name}/out"] = res
        out[f"{name}/attrs"] = np.array(repr((r.shape, r.dsize, r.fx, r.fy, r.interpolation, r.is_conservative, r._conservative_explicit)))
        assert res is not src
    try:
        da.Resize(interpolation="inter_cubic")
        out["bad_interpolation"] = np.array("no error")
    except NotImplementedError:
        out["bad_interpolation"] = np.array("NotImplementedError")
    # Images: same class and metadata back; extensive images conserve by default, an explicit choice wins
    im = da.OpticalImage(rgb.astype(np.float64), dimensions=[0.6, 0.8], name="photo")
    small = da.resize(im, shape=(30, 40), interpolation="inter_area")
    out["image/out"], out["image/meta"] = small.img, np.array([type(small).__name__, str(small.name), str(list(small.dimensions))])
    ext = da.ExtensiveImage(np.arange(1.0, 17.0).reshape(4, 4), dimensions=[1, 1], scalar=True)
    out["ext/default"] = da.Resize(shape=(2, 2))(ext).img
    out["ext/explicit_off"] = da.Resize(shape=(2, 2), **{"resize conservative": False})(ext).img
    out["ext/function"] = da.resize(ext, shape=(2, 2)).img
    # equalize_voxel_size
    aniso = da.OpticalImage(rgb.astype(np.float64), dimensions=[0.6, 1.6])
    eq = da.equalize_voxel_size(aniso)
    out["eq/default_shape"], out["eq/default_out"] = np.array(eq.img.shape), eq.img
    eq2 = da.equalize_voxel_size(da.Image(np.ones((10, 10)), width=2, height=1, scalar=True), voxel_size=0.05)
    out["eq/explicit_shape"] = np.array(eq2.img.shape)
    np.savez_compressed(os.path.join(HERE, "resize_assembly_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
