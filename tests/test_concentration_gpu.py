"""Fused ConcentrationAnalysis pipeline vs the stage-by-stage NumPy chain (bit-exact: every stage is an
elementwise expression in the image dtype; the TV part is compared with the oracle's same-dtype flow)."""
import numpy as np
import pytest

import darsia_amd as da
from oracle import concentration_numpy as OC

pytestmark = pytest.mark.gpu


def _frames(h, w, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    base = rng.random((h, w, 3)).astype(dtype) * 0.3 + 0.3
    blob = np.exp(-(((np.mgrid[0:h, 0:w][0] - h / 2) / (h / 5)) ** 2 + ((np.mgrid[0:h, 0:w][1] - w / 3) / (w / 6)) ** 2))
    probe = (base + 0.3 * blob[..., None] * np.array([1.0, 0.4, -0.2]) + rng.normal(0, 0.02, base.shape)).astype(dtype)
    return da.OpticalImage(base), da.OpticalImage(probe)


@pytest.mark.parametrize("color", ["red", "green", "blue", "red+green", "gray", "negative-key"])
@pytest.mark.parametrize("option", ["absolute", "positive", "negative", "plain"])
def test_fused_pipeline_without_restoration(color, option):
    base, probe = _frames(37, 53, 1)
    model = da.CombinedModel([da.LinearModel(scaling=2.5, offset=-0.05), da.ClipModel(min_value=0.0, max_value=1.0)])
    ca = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color=color), None, None, model, **{"diff option": option})
    out = ca(probe)
    # the checker's model stage: the reference's expressions as plain callables (linearmodel.py:140-149:
    # scaling * img + offset; clipmodel.py:87-106: np.clip -- the latter pinned by tests/golden/models_golden.npz)
    ref = OC.concentration(probe.img, base.img, option, color, None,
                           [lambda x: 2.5 * x + -0.05, lambda x: np.clip(x, 0.0, 1.0)], None, False)
    assert isinstance(out, da.ScalarImage) and out.img.dtype == ref.dtype
    assert np.array_equal(out.img, ref), np.abs(out.img - ref).max()


@pytest.mark.parametrize("restoration_first", [False, True])
def test_fused_pipeline_with_tvd_and_cleaning_filter(restoration_first):
    base, probe = _frames(64, 96, 2)
    _, noise_frame = _frames(64, 96, 3)
    noise_frame.img = (base.img + np.random.default_rng(4).normal(0, 0.01, base.img.shape)).astype(np.float32)
    tvd = da.TVD(method="heterogeneous bregman", weight=0.05, omega=1.0, regularization=0.1, max_num_iter=12, eps=None)
    model = da.CombinedModel([da.ScalingModel(scaling=3.0), da.ClipModel(min_value=0.0, max_value=1.0)])
    ca = da.ConcentrationAnalysis([base, noise_frame], da.MonochromaticReduction(color="red"), None, tvd, model,
                                  **{"diff option": "positive", "restoration -> model": restoration_first})
    assert ca.threshold_cleaning_filter is not None and ca.threshold_cleaning_filter.dtype == np.float64
    out = ca(probe)
    # float32 frames and a float64 filter: the reference subtracts and reduces in float32, the filter then widens the signal
    ref = OC.concentration(probe.img, base.img, "positive", "red",
                           ca.threshold_cleaning_filter, [lambda x: 3.0 * x, lambda x: np.clip(x, 0.0, 1.0)],
                           dict(mu=0.05, omega=1.0, ell=0.1, max_num_iter=12), restoration_first)
    assert out.img.dtype == np.float64  # the float64 cleaning filter promotes the signal, as in the reference
    assert np.array_equal(out.img, ref), np.abs(out.img - ref).max()
    # second frame re-uses the device context (base and filter stay resident)
    out2 = ca(probe)
    assert np.array_equal(out2.img, out.img)


@pytest.mark.parametrize("color", ["red", "red+green", "gray", "negative-key"])
@pytest.mark.parametrize("option", ["absolute", "positive", "negative", "plain"])
def test_float32_frames_under_a_float64_filter_subtract_in_float32(color, option):
    """NumPy's promotion, stage by stage (concentrationanalysis.py:328-395): probe - baseline and the reduction in the frames'
    own float32, only ``signal - filter`` in float64.  Frames far from their baseline (where a float32 difference rounds) tell
    this from a pipeline that widens first; the frames go to the device as float32."""
    rng = np.random.default_rng(21)
    h, w = 48, 72
    base = da.OpticalImage((rng.random((h, w, 3)) * 0.2 + 0.01).astype(np.float32))
    probe = da.OpticalImage((rng.random((h, w, 3)) * 0.9 + 0.05).astype(np.float32))
    extra = da.OpticalImage((base.img + rng.normal(0, 0.004, base.img.shape)).astype(np.float32))
    model = da.CombinedModel([da.LinearModel(scaling=1.7, offset=0.01), da.ClipModel(min_value=0.0)])
    ca = da.ConcentrationAnalysis([base, extra], da.MonochromaticReduction(color=color), None, None, model, **{"diff option": option})
    assert ca.threshold_cleaning_filter.dtype == np.float64 and ca._fusable(probe.img) is not None
    out = ca(probe)
    ref = OC.concentration(probe.img, base.img, option, color, ca.threshold_cleaning_filter,
                           [lambda x: 1.7 * x + 0.01, lambda x: np.clip(x, 0.0, None)], None, False)
    widened = OC.concentration(probe.img.astype(np.float64), base.img.astype(np.float64), option, color,
                               ca.threshold_cleaning_filter, [lambda x: 1.7 * x + 0.01, lambda x: np.clip(x, 0.0, None)], None, False)
    assert out.img.dtype == ref.dtype == np.float64 and np.array_equal(out.img, ref)
    # the data does tell the two apart (gray rounds the difference to float32 either way: cv2's conversion is float32)
    assert np.array_equal(ref, widened) == (color == "gray")
    assert list(ca.map([probe, probe]))[1].img.tobytes() == out.img.tobytes()


@pytest.mark.parametrize("dtype,scale", [(np.uint8, 255), (np.uint16, 65535)])
@pytest.mark.parametrize("base_dtype", [np.float32, np.float64])
def test_integer_frames_convert_on_the_device(dtype, scale, base_dtype):
    """Integer frames travel as they are stored and are converted on the device the way img_as(float) does
    (x * (1 / imax) in float64, concentrationanalysis.py:241-246), whatever float type the baseline has; with the restoration."""
    rng = np.random.default_rng(22)
    h, w = 40, 64
    base = da.OpticalImage((rng.random((h, w, 3)) * 0.4).astype(base_dtype))
    raw = (rng.random((h, w, 3)) * scale).astype(dtype)
    tvd = da.TVD(method="heterogeneous bregman", weight=0.05, omega=1.0, regularization=0.1, max_num_iter=6, eps=None)
    ca = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="green"), None, tvd, da.ScalingModel(scaling=2.0),
                                  **{"diff option": "absolute"})
    with pytest.warns(UserWarning, match="converted to float"):
        out = ca(da.OpticalImage(raw))
    as_float = np.multiply(raw, 1.0 / scale, dtype=np.float64)
    ref = OC.concentration(as_float, base.img, "absolute", "green", None, [lambda x: 2.0 * x],
                           dict(mu=0.05, omega=1.0, ell=0.1, max_num_iter=6), False)
    assert out.img.dtype == np.float64 and np.array_equal(out.img, ref)
    with pytest.warns(UserWarning, match="converted to float"):
        series = list(ca.map([da.OpticalImage(raw), da.OpticalImage(as_float), da.OpticalImage(raw)]))
    assert all(np.array_equal(x.img, ref) for x in series)


def test_typed_signal_upload_rejects_what_numpy_would_widen():
    from darsia_amd import _lib
    import ctypes as C

    img = np.zeros((16, 24), np.float32)
    with da.DeviceTVD(img.shape, np.float32) as st:
        st.set_coeffs(0.1, 1.0, None, img)
        lib = st.lib
        base64 = np.zeros((16, 24, 3), np.float64)
        assert lib.dxh_tvd_set_base_typed(st._h, _lib.ptr(base64), 3, _lib.DXH_F64) == -1          # float64 baseline, float32 context
        probe8 = np.zeros((16, 24, 3), np.uint8)
        assert lib.dxh_tvd_upload_signal_typed(st._h, _lib.ptr(probe8), 2, 3, 0, 1, None, 0) == -1   # uint8 -> float64 signal
        assert lib.dxh_tvd_upload_signal_typed(st._h, _lib.ptr(probe8), 7, 3, 0, 1, None, 0) == -1   # unknown dtype code
        probe32 = np.ones((16, 24, 3), np.float32)
        assert lib.dxh_tvd_upload_signal_typed(st._h, _lib.ptr(probe32), _lib.DXH_F32, 3, 0, 2, None, 0) == 0


@pytest.mark.parametrize("flow", ["default", "native"])
def test_scalar_frames_and_generic_fallback_chain(flow, monkeypatch):
    """float32 frames with a TV stage, in the flow a caller gets by default (the reference under NumPy >= 2: oracle flow
    "numpy2") and in the native float32 flow (DXH_PROMOTION=native, the reference under NumPy < 2: oracle flow "strict")."""
    if flow == "native":
        monkeypatch.setenv("DXH_PROMOTION", "native")
    oflow = "strict" if flow == "native" else "numpy2"
    rng = np.random.default_rng(5)
    base = da.ScalarImage(rng.random((40, 50)).astype(np.float32))
    probe = da.ScalarImage((base.img + 0.2 * rng.random((40, 50))).astype(np.float32))
    tvd = da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=8, eps=None)
    ca = da.ConcentrationAnalysis(base, None, None, tvd, da.ClipModel(min_value=0.05), **{"diff option": "plain"})
    out = ca(probe)
    ref = OC.concentration(probe.img, base.img, "plain", None, None, [lambda x: np.clip(x, 0.05, None)],
                           dict(mu=0.1, omega=1, ell=1.0, max_num_iter=8), False, promotion=oflow)
    assert np.array_equal(out.img, ref)
    # a callable reduction cannot be fused: the stages run one by one through the objects' __call__
    base3, probe3 = _frames(30, 44, 6)
    ca2 = da.ConcentrationAnalysis(base3, da.MonochromaticReduction(color=lambda im: im[:, :, 0] - im[:, :, 2]), None,
                                   da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=5, eps=None), None)
    out2 = ca2(probe3)
    sig = np.abs(probe3.img - base3.img)
    sig = sig[:, :, 0] - sig[:, :, 2]
    from oracle import tvd_numpy as O
    assert np.array_equal(out2.img, O.split_bregman_tvd(sig, mu=0.1, ell=1.0, max_num_iter=5, promotion=oflow))


@pytest.mark.parametrize("n_frames", [0, 1, 2, 5])
def test_series_map_equals_frame_by_frame(n_frames):
    """ConcentrationAnalysis.map pipelines the frames over two contexts / two streams; the results are those of
    calling the object frame by frame (context re-use after two frames included)."""
    base, _ = _frames(96, 128, 10)
    frames = [_frames(96, 128, 20 + i)[1] for i in range(n_frames)]
    tvd = da.TVD(method="heterogeneous bregman", weight=0.07, omega=1.0, regularization=0.2, max_num_iter=9, eps=None)
    model = da.CombinedModel([da.LinearModel(scaling=2.0, offset=0.01), da.ClipModel(min_value=0.0, max_value=1.0)])
    ca = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="gray"), None, tvd, model, **{"diff option": "absolute"})
    want = [ca(f).img for f in frames]
    got = list(ca.map(iter(frames)))
    assert len(got) == n_frames
    for g, w_ in zip(got, want):
        assert isinstance(g, da.ScalarImage) and g.img.dtype == w_.dtype and np.array_equal(g.img, w_)


def test_series_results_are_recycled_only_when_dead():
    """Large host results come out of the pool of recycled result arrays (darsia_amd/hostpool.py): a streaming caller gets the
    pages of the result it dropped, a caller who keeps results never sees one change."""
    import gc

    from darsia_amd import hostpool

    hostpool.trim()
    h, w = 1536, 1600   # 9.8 MB of float32 per result: above the pool's threshold
    base, _ = _frames(h, w, 40)
    frames = [_frames(h, w, 41 + i)[1] for i in range(4)]
    ca = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="red"), None,
                                  da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=5, eps=None), None)
    kept = [r.img for r in ca.map(frames)]
    copies = [k.copy() for k in kept]
    assert len({k.ctypes.data for k in kept}) == 4
    seen = set()
    for i, r in enumerate(ca.map(frames)):          # streaming: each result dropped before the next is fetched
        assert np.array_equal(r.img, copies[i])
        seen.add(r.img.ctypes.data)
        del r
        gc.collect()
    assert len(seen) < 4                            # pages were re-used ...
    assert not seen & {k.ctypes.data for k in kept}
    assert all(np.array_equal(k, c) for k, c in zip(kept, copies))   # ... but never those of a live result
    hostpool.trim()


def test_series_map_mixed_frames_and_early_exit():
    base, _ = _frames(48, 64, 30)
    a, b = _frames(48, 64, 31)[1], _frames(48, 64, 32)[1]
    odd = da.OpticalImage(_frames(48, 64, 33, np.float64)[1].img)   # another dtype: leaves the pipelined stretch
    ca = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="red"), None,
                                  da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=4, eps=None), None)
    want = [ca(f).img for f in (a, b, odd, a)]
    got = [g.img for g in ca.map([a, b, odd, a])]
    assert all(np.array_equal(g, w_) and g.dtype == w_.dtype for g, w_ in zip(got, want))
    # abandoning the generator releases the contexts (the in-flight upload is joined first)
    gen = ca.map([a, b, a, b])
    first = next(gen)
    gen.close()
    assert np.array_equal(first.img, want[0])
    # frames the device path cannot fuse go one by one
    ca2 = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color=lambda im: im[:, :, 1]), None, None, None)
    assert all(np.array_equal(g.img, ca2(f).img) for g, f in zip(ca2.map([a, b]), (a, b)))


@pytest.mark.parametrize("restoration_first", [False, True])
def test_default_tvd_restoration_runs_inside_the_fused_pipeline(restoration_first):
    """restoration = TVD() (the reference's default method, chambolle) between the elementwise stages, everything on the
    device: equal to the stage-by-stage chain -- NumPy stages of the checker around the device operator on a host
    array -- and the series pipeline gives the same frames."""
    base, probe = _frames(72, 100, 40)
    tvd = da.TVD(weight=0.08, max_num_iter=30, eps=1e-4)
    assert tvd.method == "chambolle"
    model = da.CombinedModel([da.LinearModel(scaling=1.7, offset=0.02), da.ClipModel(min_value=0.0, max_value=1.0)])
    ca = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="red+green"), None, tvd, model,
                                  **{"diff option": "absolute", "restoration -> model": restoration_first})
    assert ca._fusable(probe.img) is not None
    out = ca(probe)
    signal = OC.concentration(probe.img, base.img, "absolute", "red+green", None, [], None, False)   # reduce only
    stages = [lambda x: 1.7 * x + 0.02, lambda x: np.clip(x, 0.0, 1.0)]
    apply_model = lambda x: stages[1](stages[0](x))
    want = apply_model(da.denoise_tv_chambolle(signal, weight=0.08, max_num_iter=30, eps=1e-4)) if restoration_first \
        else da.denoise_tv_chambolle(apply_model(signal), weight=0.08, max_num_iter=30, eps=1e-4)
    assert out.img.dtype == want.dtype and np.array_equal(out.img, want)
    frames = [probe, _frames(72, 100, 41)[1], _frames(72, 100, 42)[1]]
    series = list(ca.map(frames))
    assert all(np.array_equal(s.img, ca(f).img) for s, f in zip(series, frames))


@pytest.mark.parametrize("method", ["anisotropic bregman", "isotropic bregman"])
def test_gauss_seidel_restorations_stay_on_the_device(method):
    """scikit-image's split-Bregman methods take and return device arrays (same bits as the host call), which lets
    ConcentrationAnalysis keep the frame on the device around them."""
    rng = np.random.default_rng(50)
    sig = rng.random((60, 84)).astype(np.float32)
    op = da.TVD(method=method, weight=0.4, max_num_iter=12, eps=1e-9)
    want = op(sig)
    dev = op(da.DeviceArray.from_host(sig))
    assert isinstance(dev, da.DeviceArray) and np.array_equal(dev.get(), want)
    base, probe = _frames(60, 84, 51)
    ca = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="blue"), None, op, da.ClipModel(min_value=0.0),
                                  **{"diff option": "plain"})
    assert ca._fusable(probe.img) is not None
    signal = OC.concentration(probe.img, base.img, "plain", "blue", None, [], None, False)
    assert np.array_equal(ca(probe).img, op(np.clip(np.ascontiguousarray(signal), 0.0, None)))


def test_weight_map_stays_in_the_context_between_frames_and_edits_are_noticed():
    """Heterogeneous TVD with a porosity-like weight map inside the fused pipeline: the coefficients are set once per
    context, a map edited in place (or replaced) is uploaded again."""
    base, probe = _frames(64, 96, 60)
    rng = np.random.default_rng(61)
    # (float32 like the signal: a float64 map next to float32 frames runs the operator's numpy2 flow instead of the
    # context's own iterations, test_float32_signal_with_float64_weights_... below)
    weight = (0.02 + 0.1 * rng.random((64, 96))).astype(np.float32)
    tvd = da.TVD(method="heterogeneous bregman", weight=weight, omega=1.0, regularization=0.3, max_num_iter=7, eps=None)
    ca = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="green"), None, tvd, None, **{"diff option": "positive"})

    def reference(w):
        sig = OC.concentration(probe.img, base.img, "positive", "green", None, [], None, False)
        return da.split_bregman_tvd(np.ascontiguousarray(sig), mu=w.astype(np.float32), omega=1.0, ell=0.3, max_num_iter=7)

    first = ca(probe).img
    assert np.array_equal(first, reference(weight))
    assert np.array_equal(ca(probe).img, first)                  # second frame: coefficients already in the context
    weight[10:40, 20:70] *= 3.0                                  # in-place edit of the map the operator holds
    assert np.array_equal(ca(probe).img, reference(weight))
    tvd.weight = 0.5 * weight                                    # another map
    assert np.array_equal(ca(probe).img, reference(0.5 * weight))


def test_integer_frames_are_converted_like_the_reference():
    """concentrationanalysis.py:241-246: a non-float probing image is converted with img_as(float) and a warning."""
    base, probe = _frames(40, 56, 70, np.float64)
    u8 = da.OpticalImage((np.clip(probe.img, 0, 1) * 255).astype(np.uint8))
    ca = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="red"), None, None, None, **{"diff option": "plain"})
    with pytest.warns(UserWarning, match="converted to float"):
        out = ca(u8)
    want = np.multiply(u8.img[..., 0], 1.0 / 255, dtype=np.float64) - base.img[..., 0]   # scikit-image multiplies by 1 / 255
    assert out.img.dtype == np.float64 and np.array_equal(out.img, want)
    with pytest.raises(TypeError):
        ca(da.OpticalImage(np.zeros((40, 56, 3), dtype=np.int32)))


def _golden_cases():
    import ast
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "concentration_golden.npz"))
    return g, sorted({k.split("/")[0] for k in g.files if k.endswith("/spec")}), ast


@pytest.mark.parametrize("tag", _golden_cases()[1])
def test_product_against_outputs_of_the_reference_class(tag, monkeypatch):
    """The fused device pipeline against what the reference's own ConcentrationAnalysis returned for the same inputs
    (tests/golden/make_golden_concentration.py): float64 bit for bit; float32 with TVD bit for bit as well -- the vectors
    were made under NumPy 1.26 (the only NumPy next to a real scikit-image here), whose value-based casting keeps the
    reference's iteration in float32: that vector is the NATIVE flow's (DXH_PROMOTION=native); the default flow of the same
    configuration is checked against the oracle's NumPy-2 flow below."""
    g, _, ast = _golden_cases()
    spec = ast.literal_eval(str(g[f"{tag}/spec"]))
    float32_tv = spec["rest"] is not None and g[f"{tag}/probe"].dtype == np.float32
    if float32_tv:
        monkeypatch.setenv("DXH_PROMOTION", "native")
    base, probe = g[f"{tag}/base"], g[f"{tag}/probe"]
    extras = [g[f"{tag}/extra{i}"] for i in range(spec["nclean"])]
    model = {"linear_clip": lambda: da.CombinedModel([da.LinearModel(scaling=2.5, offset=-0.05), da.ClipModel(min_value=0.0, max_value=1.0)]),
             "clip_only": lambda: da.ClipModel(min_value=0.02), "scaling": lambda: da.ScalingModel(scaling=3.0)}[spec["model"]]()
    rest = None
    if spec["rest"] is not None:
        rest = da.TVD(method="heterogeneous bregman", eps=None, **spec["rest"])
    bases = [da.OpticalImage(base)] + [da.OpticalImage(e) for e in extras]
    ca = da.ConcentrationAnalysis(bases if extras else bases[0], da.MonochromaticReduction(color=spec["color"]), None, rest, model,
                                  **{"diff option": spec["diff"], "restoration -> model": spec["rest_first"]})
    assert ca._fusable(probe) is not None
    out = ca(da.OpticalImage(probe, name="frame"))
    want = g[f"{tag}/out"]
    assert isinstance(out, da.ScalarImage) and out.name == "frame" and out.img.dtype == want.dtype
    assert np.array_equal(out.img, want), np.abs(out.img - want).max()
    if float32_tv:
        # the same configuration in the product's default flow: the reference's arithmetic under NumPy >= 2
        assert extras == [] and spec["model"] == "scaling" and not spec["rest_first"]
        monkeypatch.delenv("DXH_PROMOTION")
        ca = da.ConcentrationAnalysis(bases[0], da.MonochromaticReduction(color=spec["color"]), None,
                                      da.TVD(method="heterogeneous bregman", eps=None, **spec["rest"]), model,
                                      **{"diff option": spec["diff"], "restoration -> model": spec["rest_first"]})
        out = ca(da.OpticalImage(probe, name="frame"))
        r = spec["rest"]
        ref = OC.concentration(probe, base, spec["diff"], spec["color"], None, [lambda x: 3.0 * x],
                               dict(mu=r["weight"], omega=r.get("omega", 1.0), ell=r.get("regularization", 1.0),
                                    max_num_iter=r["max_num_iter"]), False, promotion="numpy2")
        assert out.img.dtype == ref.dtype and np.array_equal(out.img, ref)
        assert not np.array_equal(ref, want)    # the two NumPy generations of the reference do differ on this input


def test_device_resident_frames_stay_on_the_device():
    """A frame that is in HBM already (out of the colour correction, say) is read where it lies and the concentration comes back
    as a device array: same bits as the host-array call."""
    rng = np.random.default_rng(31)
    h, w = 60, 88
    base = da.OpticalImage((rng.random((h, w, 3)) * 0.5).astype(np.float32))
    raw = (rng.random((h, w, 3)) * 255).astype(np.uint8)
    balance = da.AffineBalance()
    balance.balance_scaling, balance.balance_translation = np.eye(3) + 0.05 * rng.random((3, 3)), 0.01 * rng.random(3)
    tvd = da.TVD(method="heterogeneous bregman", weight=0.05, omega=1.0, regularization=0.1, max_num_iter=8, eps=None)
    model = da.CombinedModel([da.LinearModel(scaling=2.0, offset=0.0), da.ClipModel(min_value=0.0, max_value=1.0)])
    ca = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="red"), None, tvd, model, **{"diff option": "positive"})
    corrected_host = balance.apply_balance(raw)                                   # float32 on the host
    want = ca(da.OpticalImage(corrected_host))
    d_raw = da.DeviceArray.from_host(raw)
    d_corrected = balance.apply_balance(d_raw)                                    # stays in HBM
    assert isinstance(d_corrected, da.DeviceArray)
    got = ca(da.OpticalImage(d_corrected, name="frame"))
    assert isinstance(got, da.ScalarImage) and isinstance(got.img, da.DeviceArray) and got.name == "frame"
    assert np.array_equal(got.img.get(), want.img)
    series = list(ca.map([da.OpticalImage(d_corrected), da.OpticalImage(corrected_host), da.OpticalImage(d_corrected)]))
    assert [type(x.img).__name__ for x in series] == ["DeviceArray", "ndarray", "DeviceArray"]
    assert all(np.array_equal(x.img.get() if isinstance(x.img, da.DeviceArray) else x.img, want.img) for x in series)
    dup = got.copy()                                                              # Image.copy(): a second device array
    assert dup.img.ptr != got.img.ptr and np.array_equal(dup.img.get(), want.img)
    generic = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color=lambda im: im[..., 0]), None, None, model)
    with pytest.raises(NotImplementedError, match="fused pipeline"):
        generic(da.OpticalImage(d_corrected))


def test_random_sweep_of_frame_types_against_numpy_promotion():
    """Sixty random combinations of frame type (float32 / float64 / uint8 / uint16), baseline (none / float32 / float64), cleaning
    filter, difference option, reduction and shape: the fused device pipeline against the stage-by-stage NumPy chain, whose
    dtypes are whatever NumPy's promotion makes them (integer frames first converted the img_as(float) way)."""
    import warnings

    rng = np.random.default_rng(2024)
    colors = ["red", "green", "blue", "red+green", "negative-key", "gray"]
    for case in range(60):
        h, w = int(rng.integers(9, 70)), int(rng.integers(9, 90))
        frame_dt = [np.float32, np.float64, np.uint8, np.uint16][int(rng.integers(4))]
        base_dt = [None, np.float32, np.float64][int(rng.integers(3))]
        with_filter = base_dt is not None and bool(rng.integers(2))
        option = ["absolute", "positive", "negative", "plain"][int(rng.integers(4))]
        color = colors[int(rng.integers(len(colors)))]
        vals = rng.random((h, w, 3)) * 1.2 - 0.1 if np.dtype(frame_dt).kind == "f" else rng.random((h, w, 3))
        frame = vals.astype(frame_dt) if np.dtype(frame_dt).kind == "f" else (vals * np.iinfo(frame_dt).max).astype(frame_dt)
        as_float = frame if frame.dtype.kind == "f" else np.multiply(frame, 1.0 / np.iinfo(frame_dt).max, dtype=np.float64)
        base = None if base_dt is None else (rng.random((h, w, 3)) * 0.5).astype(base_dt)
        wide = as_float.dtype if base is None else np.result_type(as_float.dtype, base.dtype)
        if color == "gray" and wide != np.float32:
            color = "red"                       # cv2's gray conversion is float32-only: the product leaves the fused path there
        bases = None if base is None else da.OpticalImage(base)
        if with_filter:
            bases = [da.OpticalImage(base), da.OpticalImage((base + rng.normal(0, 0.01, base.shape)).astype(base_dt))]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ca = da.ConcentrationAnalysis(bases, da.MonochromaticReduction(color=color), None, None,
                                          da.LinearModel(scaling=1.5, offset=-0.02), **{"diff option": option})
            assert ca._fusable(frame) is not None, (case, frame_dt, base_dt, color)
            out = ca(da.OpticalImage(frame))
        ref = OC.concentration(as_float, base, option, color, ca.threshold_cleaning_filter, [lambda x: 1.5 * x + -0.02], None, False)
        assert out.img.dtype == ref.dtype, (case, frame_dt, base_dt, with_filter, out.img.dtype, ref.dtype)
        assert np.array_equal(out.img, ref), (case, frame_dt, base_dt, with_filter, option, color)
        ca.close()


def test_float32_signal_with_float64_weights_takes_the_reference_flow_and_paths_are_observable(caplog, monkeypatch):
    """(Runs in the product's DEFAULT dtype flow: this module otherwise pins the native float32 kernels, conftest.py.)
    The workflow's default (presets/workflows/restoration.py:117-148): float32 frames, float64 porosity weights.  The TVD
    operator alone runs the reference's mixed dtype flow there (promotion "numpy2"); inside ConcentrationAnalysis the SAME
    operator runs on the device-resident signal between the two elementwise stages, so the fused object and the operator
    called by hand give the same bits -- which are the oracle's reference-flow bits.  ``last_path`` says which pipeline
    served a frame; the host chain also logs why."""
    import logging

    from oracle import tvd_numpy as O

    monkeypatch.delenv("DXH_PROMOTION", raising=False)
    base, probe = _frames(48, 72, 11)
    rng = np.random.default_rng(5)
    weight = 0.08 * (0.3 + 0.7 * rng.random((48, 72)))          # float64
    weight[10:20, 30:50] = 0.0
    tvd = da.TVD(method="heterogeneous bregman", weight=weight, omega=1.0, regularization=1.0, max_num_iter=25, eps=None)
    ca = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="red"), None, tvd, None, **{"diff option": "positive"})
    out = ca(probe)
    assert ca.last_path == "fused" and ca.last_path_reason is None and out.img.dtype == np.float32
    signal = OC.concentration(probe.img, base.img, "positive", "red", None, [], None, False)
    assert signal.dtype == np.float32
    want = O.split_bregman_tvd(signal, mu=weight, omega=1.0, ell=1.0, max_num_iter=25)      # the reference's NumPy-2 flow
    assert np.array_equal(out.img, want)
    assert np.array_equal(tvd(signal), want)
    # promotion="native": the context's own float32 iterations (bit-exact against the oracle's same-dtype flow)
    tvd32 = da.TVD(method="heterogeneous bregman", weight=weight.astype(np.float32), omega=1.0, regularization=1.0, max_num_iter=25, eps=None,
                   promotion="native")
    ca32 = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="red"), None, tvd32, None, **{"diff option": "positive"})
    got32 = ca32(probe).img
    assert ca32.last_path == "fused"
    assert np.array_equal(got32, O.split_bregman_tvd(signal, mu=weight.astype(np.float32), omega=1.0, ell=1.0, max_num_iter=25, promotion="strict"))
    np.testing.assert_allclose(got32, want, rtol=1e-4, atol=1e-6)
    # a stage the device pipeline does not have: the host chain serves the frame, says so, and gives the same answer
    # as the stages applied by hand
    ca_gen = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="red"), lambda x: x, tvd, None, **{"diff option": "positive"})
    with caplog.at_level(logging.INFO, logger="darsia_amd.concentration"):
        out_gen = ca_gen(probe)
    assert ca_gen.last_path == "generic" and "balancing" in ca_gen.last_path_reason
    assert any("stage-by-stage" in r.message for r in caplog.records)
    assert np.array_equal(out_gen.img, want)
    # refresh(): an in-place edit of the weight map is outside what the device caches can see; after refresh() it counts
    weight32 = weight.astype(np.float32)
    tvd_e = da.TVD(method="heterogeneous bregman", weight=weight32, omega=1.0, regularization=1.0, max_num_iter=8, eps=None,
                   promotion="native")
    ca_e = da.ConcentrationAnalysis(base, da.MonochromaticReduction(color="red"), None, tvd_e, None, **{"diff option": "positive"})
    first = ca_e(probe).img.copy()
    weight32[25, 37] *= 3.0                 # one pixel: no sample of the fingerprint sees it
    ca_e.refresh()
    second = ca_e(probe).img
    assert not np.array_equal(first, second)
    assert np.array_equal(second, O.split_bregman_tvd(signal, mu=weight32, omega=1.0, ell=1.0, max_num_iter=8, promotion="strict"))
