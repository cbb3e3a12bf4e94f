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
TAGS = sorted({k.split("/")[0] for k in G.files if k.endswith("/out")})


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
