"""Golden vectors for the ASSEMBLY of the piecewise perspective warp (corrections/shape/piecewiseperspective.py:35-145):
the reference's own ``PiecewisePerspectiveTransform.find_and_warp`` and ``darsia.Patches`` are executed on the reference's own
``Image`` class -- patch loop, corner displacement, effective origin / size of every warped tile, clipping to the canvas,
additive seams, cast back to the input dtype, metadata of the result.

The two OpenCV functions that loop calls, ``cv2.getPerspectiveTransform`` and ``cv2.warpPerspective(..., INTER_LINEAR)``, do
not exist here.  They are provided by a ``cv2`` module backed by THIS repository's restatement of OpenCV's published
algorithms (oracle/warp_numpy.py: get_perspective_transform, warp_perspective).  These vectors therefore pin everything
around those two calls on the reference's code, and nothing about the two calls themselves, whose parity with OpenCV
stays unpinned (oracle/warp_numpy.py header).

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
    np.savez_compressed(os.path.join(HERE, "warp_assembly_golden.npz"), **out)
    print("wrote", len(out), "arrays;", {k: v.dtype.name for k, v in out.items() if k.endswith("/out")})


if __name__ == "__main__":
    main()
