"""oracle/curvature_numpy.py against outputs of the reference's own CurvatureCorrection (tests/golden/make_golden_curvature.py
executes corrections/shape/curvature.py and basecorrection.py): cached coordinate grid and corrected image, orders 0 / 1,
four dtypes, scalar and colour images, resize_factor."""
import ast
import copy
import os

import numpy as np
import pytest

from oracle import curvature_numpy as cu

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "curvature_golden.npz"))
TAGS = sorted({k.split("/")[0] for k in G.files if k.endswith("/spec")})
CONFIGS = {
    "bulge": {"bulge": dict(horizontal_bulge=5e-6, vertical_bulge=-3e-6)},
    "all3": {"init": dict(horizontal_bulge=4e-7, vertical_bulge=1e-7),
             "bulge": dict(horizontal_bulge=-2e-7, vertical_bulge=3e-7, horizontal_center_offset=4),
             "stretch": dict(horizontal_stretch=2e-7, vertical_stretch=-1e-7, vertical_center_offset=2)},
    "stretch": {"stretch": dict(horizontal_stretch=3e-6, vertical_stretch=2e-6, horizontal_center_offset=-3, vertical_center_offset=5)},
}


def scaled_config(name, rf):
    """The reference adapts the configuration to resize_factor (curvature.py:827-855)."""
    cfg = copy.deepcopy(CONFIGS[name])
    for main in ("init", "bulge"):
        for key in ("horizontal_bulge", "vertical_bulge", "horizontal_center_offset", "vertical_center_offset"):
            if main in cfg and key in cfg[main]:
                cfg[main][key] *= rf
    if "stretch" in cfg:
        for key in ("horizontal_stretch", "vertical_stretch", "horizontal_center_offset", "vertical_center_offset"):
            if key in cfg["stretch"]:
                cfg["stretch"][key] *= rf
    return cfg


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_reproduces_the_reference_class(tag):
    spec = ast.literal_eval(str(G[f"{tag}/spec"]))
    img = G[f"{tag}/img"]
    cfg = scaled_config(spec["config"], spec["resize_factor"]) if spec["resize_factor"] != 1.0 else CONFIGS[spec["config"]]
    grid, shape = cu.precompute_grid(img.shape, cfg, order=spec["order"])   # the coordinate images go through the same gather
    assert tuple(shape) == tuple(G[f"{tag}/shape"])
    assert grid.dtype == G[f"{tag}/grid"].dtype and np.array_equal(grid, G[f"{tag}/grid"])
    out = cu.transform_image(img, grid, shape, order=spec["order"])
    want = G[f"{tag}/out"]
    assert out.dtype == want.dtype and np.array_equal(out, want)
