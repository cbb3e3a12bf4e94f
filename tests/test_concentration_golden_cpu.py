"""The checker of the ConcentrationAnalysis tests (oracle/concentration_numpy.py) and the product's host-side set-up
against outputs of the reference's own class (tests/golden/make_golden_concentration.py: the reference modules executed
under scikit-image 0.18.3 / NumPy 1.26): all four difference options, five colour reductions, cleaning filters from extra
baselines, Linear / Clip / Scaling / Combined models, TVD("heterogeneous bregman") before and after the model."""
import ast
import os

import numpy as np
import pytest

import darsia_amd as da
from oracle import concentration_numpy as OC

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "concentration_golden.npz"))
TAGS = sorted({k.split("/")[0] for k in G.files if k.endswith("/spec")})
MODELS = {"linear_clip": [lambda x: 2.5 * x + -0.05, lambda x: np.clip(x, 0.0, 1.0)],
          "clip_only": [lambda x: np.clip(x, 0.02, None)], "scaling": [lambda x: 3.0 * x]}


def _case(tag):
    spec = ast.literal_eval(str(G[f"{tag}/spec"]))
    extras = [G[f"{tag}/extra{i}"] for i in range(spec["nclean"])]
    return spec, G[f"{tag}/base"], G[f"{tag}/probe"], extras, G[f"{tag}/out"]


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_reproduces_the_reference_class(tag):
    spec, base, probe, extras, want = _case(tag)
    filt = G[f"{tag}/filter"] if spec["nclean"] else None
    rest = None
    if spec["rest"] is not None:
        r = spec["rest"]
        rest = dict(mu=r["weight"], omega=r["omega"], ell=1.0, max_num_iter=r["max_num_iter"])   # TVD's regularization default
    got = OC.concentration(probe, base, spec["diff"], spec["color"], filt, MODELS[spec["model"]], rest, spec["rest_first"])
    assert got.dtype == want.dtype and got.shape == want.shape
    assert np.array_equal(got, want), np.abs(got - want).max()


@pytest.mark.parametrize("tag", [t for t in TAGS if "clean" in t])
def test_cleaning_filter_of_the_product_equals_the_reference(tag):
    """find_cleaning_filter (concentrationanalysis.py:154-197) is host work in the product as well."""
    spec, base, probe, extras, _ = _case(tag)
    ca = da.ConcentrationAnalysis([da.OpticalImage(base)] + [da.OpticalImage(e) for e in extras],
                                  da.MonochromaticReduction(color=spec["color"]), None, None, None, **{"diff option": spec["diff"]})
    assert ca.threshold_cleaning_filter.dtype == G[f"{tag}/filter"].dtype
    assert np.array_equal(ca.threshold_cleaning_filter, G[f"{tag}/filter"])


def test_integer_frame_vector():
    assert bool(G["u8/warned"])
    u8, base = G["u8/probe"], G["u8/base"]
    want = np.multiply(u8[..., 0], 1.0 / 255, dtype=np.float64) - base[..., 0]
    assert np.array_equal(G["u8/out"], want)      # what tests/test_concentration_gpu.py expects of the product
