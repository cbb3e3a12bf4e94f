"""The restoration configuration contract and factory (SURVEY.md section 8b "config contract"): the behaviour the
reference pins in tests/unit/test_restoration_config.py and tests/unit/test_restoration_workflow.py, re-expressed against
``darsia_amd.presets.workflows``."""
from types import SimpleNamespace

import numpy as np
import pytest

import darsia_amd as da
from darsia_amd.presets.workflows import build_restoration
from darsia_amd.presets.workflows.config import RestorationConfig, TVDConfig, VolumeAveragingConfig


def _toml(tmp_path, content):
    p = tmp_path / "config.toml"
    p.write_text(content)
    return p


def test_volume_average_defaults_and_custom_rev_size(tmp_path):
    cfg = RestorationConfig().load(_toml(tmp_path, '[restoration]\nmethod = "volume_average"\n'))
    assert cfg.method == "volume_average" and isinstance(cfg.options, VolumeAveragingConfig) and cfg.options.rev_size == 3
    cfg = RestorationConfig().load(_toml(tmp_path, '[restoration]\nmethod = "volume_average"\n\n[restoration.options]\nrev_size = 5\n'))
    assert cfg.options.rev_size == 5


def test_tvd_defaults(tmp_path):
    cfg = RestorationConfig().load(_toml(tmp_path, '[restoration]\nmethod = "tvd"\n'))
    o = cfg.options
    assert cfg.method == "tvd" and isinstance(o, TVDConfig)
    assert (o.method, o.max_num_iter, o.kwargs) == ("chambolle", 200, {})
    assert o.weight == pytest.approx(0.1) and o.eps == pytest.approx(2e-4)
    assert o.omega == pytest.approx(1.0) and o.regularization == pytest.approx(1.0)


def test_tvd_custom_options_and_extra_kwargs(tmp_path):
    cfg = RestorationConfig().load(_toml(tmp_path, (
        '[restoration]\nmethod = "TVD"\nignore = ["boolean_porosity"]\n\n[restoration.options]\n'
        'method = "isotropic bregman"\nweight = 0.05\nmax_num_iter = 100\neps = 1e-3\nomega = 2.0\nregularization = 0.5\n'
        'dim = 2\n')))
    o = cfg.options
    assert cfg.method == "tvd" and cfg.ignore == ["boolean_porosity"]
    assert (o.method, o.max_num_iter) == ("isotropic bregman", 100)
    assert o.weight == pytest.approx(0.05) and o.eps == pytest.approx(1e-3)
    assert o.omega == pytest.approx(2.0) and o.regularization == pytest.approx(0.5)
    assert o.kwargs == {"dim": 2}            # unknown keys travel on to the operator


@pytest.mark.parametrize("name", ["image_porosity", "boolean_porosity"])
def test_tvd_porosity_weight_strings(tmp_path, name):
    cfg = RestorationConfig().load(_toml(tmp_path, f'[restoration]\nmethod = "tvd"\n\n[restoration.options]\nweight = "{name}"\n'))
    assert cfg.options.weight == name


def test_method_none_unknown_and_bad_ignore(tmp_path):
    assert RestorationConfig().load(_toml(tmp_path, '[restoration]\nmethod = "none"\n')).method is None
    with pytest.raises(NotImplementedError):
        RestorationConfig().load(_toml(tmp_path, '[restoration]\nmethod = "unknown_method"\n'))
    with pytest.raises(ValueError):
        RestorationConfig().load({"method": "tvd", "ignore": [1, 2]})
    with pytest.raises(KeyError):
        RestorationConfig().load({})
    # a dict is taken as the section itself
    assert RestorationConfig().load({"method": "tvd", "options": {"weight": 3}}).options.weight == 3.0


def _rig(boolean):
    return SimpleNamespace(
        baseline=da.ScalarImage(np.zeros((2, 2), dtype=float), space_dim=2),
        image_porosity=da.ScalarImage(np.array([[0.2, 0.4], [0.6, 0.8]], dtype=float), space_dim=2),
        boolean_porosity=da.ScalarImage(np.array(boolean, dtype=bool), space_dim=2))


def test_ignore_mask_is_applied_to_the_volume_averaging_mask():
    captured = {}

    class FakeVolumeAveraging:
        def __init__(self, rev, mask):
            captured["mask"], captured["rev"] = mask, rev

    backend = SimpleNamespace(VolumeAveraging=FakeVolumeAveraging, REV=lambda size, img: SimpleNamespace(size=size, img=img))
    cfg = RestorationConfig(method="volume_average", options=VolumeAveragingConfig(rev_size=3), ignore=["boolean_porosity"])
    rig = _rig([[True, False], [True, True]])
    build_restoration(cfg, rig, backend=backend)
    np.testing.assert_allclose(captured["mask"].img, np.array([[0.2, 0.0], [0.6, 0.8]]))
    assert captured["rev"].size == 3
    assert rig.image_porosity.img[0, 1] == 0.4                  # the rig's own porosity image is left alone
    with pytest.raises(NotImplementedError):                     # no volume averaging in the TV package itself
        build_restoration(cfg, rig)


def test_ignore_mask_turns_a_scalar_tvd_weight_into_a_heterogeneous_one():
    cfg = RestorationConfig(method="tvd", options=TVDConfig(method="chambolle", weight=2.0), ignore=["boolean_porosity"])
    op = build_restoration(cfg, _rig([[True, False], [False, True]]))
    assert isinstance(op, da.TVD) and op.method == "heterogeneous bregman"
    np.testing.assert_allclose(op.weight, np.array([[2.0, 0.0], [0.0, 2.0]]))
    # Q4: the reference's operator would fail on the forwarded 'regularization'; here it is consumed
    assert op.regularization == 1.0 and "regularization" not in op.kwargs


def test_porosity_weight_string_selects_the_heterogeneous_method_and_unknowns_raise():
    rig = _rig([[True, True], [True, True]])
    op = build_restoration(RestorationConfig(method="tvd", options=TVDConfig(weight="image_porosity", omega=2.0,
                                                                             regularization=0.5, max_num_iter=7, eps=1e-3)), rig)
    assert op.method == "heterogeneous bregman" and np.array_equal(op.weight, rig.image_porosity.img)
    assert (op.omega, op.regularization, op.max_num_iter, op.eps) == (2.0, 0.5, 7, 1e-3)
    plain = build_restoration(RestorationConfig(method="tvd", options=TVDConfig(weight=0.3)), rig)
    assert plain.method == "chambolle" and plain.weight == 0.3
    with pytest.raises(ValueError, match="Unknown weight string"):
        build_restoration(RestorationConfig(method="tvd", options=TVDConfig(weight="porosity")), rig)
    with pytest.raises(ValueError, match="Unknown restoration ignore mask"):
        build_restoration(RestorationConfig(method="tvd", options=TVDConfig(weight=0.1), ignore=["unknown-mask"]), rig)
    assert build_restoration(None, rig) is None and build_restoration(RestorationConfig(method=None), rig) is None
