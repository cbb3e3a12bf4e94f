"""Host logic of round 3 that needs no GPU: which dtype flow a split_bregman_tvd call takes (choose_promotion /
numpy2_unsupported), what keeps a ConcentrationAnalysis frame off the fused device pipeline (and that it says so), the
estimator's acceptance rule in its NumPy statement, the per-communicator rendezvous file names."""
import numpy as np
import pytest

import darsia_amd as da
from darsia_amd.restoration.split_bregman_tvd import choose_promotion, numpy2_unsupported




def _why(img, mu=0.1, omega=1.0, ell=1.0, dim=2, x0=None, iso=False, solver=None, adaptive=None, quiet=True):
    return numpy2_unsupported(img, mu, omega, ell, dim, x0, iso, solver or da.Jacobi(), adaptive, quiet)


def test_where_the_numpy2_flow_exists():
    f32 = np.zeros((8, 9), np.float32)
    assert _why(f32) is None
    assert _why(np.zeros((8, 9, 3), np.float32), mu=np.ones((8, 9, 3))) is None
    assert "float32" in _why(np.zeros((8, 9)))                                   # float64 images run natively (and exactly)
    assert _why(f32, omega=np.ones((8, 9))) is None                              # float64 arrays: k_rhs_np2_het
    assert _why(f32, ell=np.ones((8, 9))) is None
    assert _why(f32, mu=np.ones((8, 9)), ell=None) is None                       # ell = 2 mu is a float64 array
    assert "float64 arrays only" in _why(f32, mu=np.ones((8, 9)), ell=np.ones((8, 9), np.float32))
    from darsia_amd.restoration.split_bregman_tvd import native_is_the_reference_flow as nat
    assert nat(f32, np.ones((8, 9), np.float32), 1.0, None) and nat(f32, 0.1, np.ones((8, 9), np.float32), 1.0)
    assert not nat(f32, np.ones((8, 9), np.float32), 1.0, 1.0)                   # scalar ell: np.float64 diag_scaled
    assert not nat(f32, np.ones((8, 9)), 1.0, None) and not nat(np.zeros((8, 9)), np.ones((8, 9), np.float32), 1.0, None)
    assert "anisotropic" in _why(f32, iso=True)
    assert _why(f32, solver=da.Jacobi(maxiter=3)) is None                         # several sweeps: three kinds of launches
    assert _why(f32, solver=da.Jacobi(maxiter=9, tol=1e-3)) is None
    assert "Jacobi" in _why(f32, solver=da.MG(2))
    assert "warm start" in _why(f32, x0=(f32, None, None))
    assert _why(np.zeros((4, 5, 6), np.float32), dim=3) is None                  # volumes
    assert _why(np.zeros((4, 5, 6), np.float32)) is None                         # six "channels"
    assert "2 or 3 axes" in _why(np.zeros((4, 5, 6, 2), np.float32), dim=3)
    assert "2 or 3 axes" in _why(f32, dim=3)


def test_choice_of_the_dtype_flow(monkeypatch):
    monkeypatch.delenv("DXH_PROMOTION", raising=False)
    mu64, mu32 = np.ones((4, 4)), np.ones((4, 4), np.float32)
    # default: the reference's own flow for every float32 image whose configuration has it
    assert choose_promotion(None, np.float32, (mu64, 1.0, 1.0), None) == "numpy2"
    assert choose_promotion(None, np.float32, (mu32, 1.0, 1.0), None) == "numpy2"
    assert choose_promotion(None, np.float32, (0.1, 1.0, None), None) == "numpy2"
    assert choose_promotion(None, np.float64, (mu64, 1.0, 1.0), "the image is not float32") == "native"
    assert choose_promotion(None, np.float32, (mu64, 1.0, None), "an array-valued mu needs a scalar ell") == "native"
    # explicit wins; an explicit numpy2 where the flow does not exist is the caller's error (raised by split_bregman_tvd)
    assert choose_promotion("native", np.float32, (mu64, 1.0, 1.0), None) == "native"
    assert choose_promotion("numpy2", np.float32, (0.1, 1.0, None), "anisotropic only") == "numpy2"
    with pytest.raises(ValueError):
        choose_promotion("float16", np.float32, (0.1, 1.0, None), None)
    # the environment variable sets the default: native everywhere, or numpy2 wherever it exists
    monkeypatch.setenv("DXH_PROMOTION", "native")
    assert choose_promotion(None, np.float32, (mu64, 1.0, 1.0), None) == "native"
    monkeypatch.setenv("DXH_PROMOTION", "numpy2")
    assert choose_promotion(None, np.float32, (0.1, 1.0, None), None) == "numpy2"
    assert choose_promotion(None, np.float32, (0.1, 1.0, None), "anisotropic only") == "native"


def test_concentration_analysis_says_why_a_frame_leaves_the_fused_pipeline():
    rng = np.random.default_rng(0)
    base = da.OpticalImage(rng.random((12, 14, 3)).astype(np.float32))
    probe = rng.random((12, 14, 3)).astype(np.float32)
    red = da.MonochromaticReduction(color="red")
    plan, why = da.ConcentrationAnalysis(base, red, None, None, None)._fusable_why(probe)
    assert plan is not None and why is None
    for kwargs, word in ((dict(balancing=lambda x: x), "balancing"),
                         (dict(restoration=lambda x: x), "not a TVD operator"),
                         (dict(restoration=da.TVD(method="heterogeneous bregman", dim=3)), "keyword arguments"),
                         (dict(restoration=da.TVD(method="heterogeneous bregman", weight=np.ones((3, 3)))), "another shape"),
                         (dict(model=lambda x: x), "no device stage")):
        args = dict(signal_reduction=red, balancing=None, restoration=None, model=None)
        args.update(kwargs)
        ca = da.ConcentrationAnalysis(base, args["signal_reduction"], args["balancing"], args["restoration"], args["model"])
        plan, why = ca._fusable_why(probe)
        assert plan is None and word in why, (word, why)
    # float32 signal with float64 weights: the pipeline's scalar context is a numpy2 context then (mode "hb2")
    tvd = da.TVD(method="heterogeneous bregman", weight=np.full((12, 14), 0.1), max_num_iter=3, eps=None)
    plan, why = da.ConcentrationAnalysis(base, red, None, tvd, None)._fusable_why(probe)
    assert why is None and plan[-1] == "hb2"
    tvd32 = da.TVD(method="heterogeneous bregman", weight=np.full((12, 14), 0.1, np.float32), max_num_iter=3, eps=None)
    plan, why = da.ConcentrationAnalysis(base, red, None, tvd32, None)._fusable_why(probe)
    assert why is None and plan[-1] == "hb2"                    # float32 signal: the reference's flow whatever the weights' type
    tvd_native = da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=3, eps=None, promotion="native")
    plan, why = da.ConcentrationAnalysis(base, red, None, tvd_native, None)._fusable_why(probe)
    assert why is None and plan[-1] == "hb"                     # asked for: the context's own float32 iterations
    base64 = da.OpticalImage(base.img.astype(np.float64))
    plan, why = da.ConcentrationAnalysis(base64, red, None, tvd32, None)._fusable_why(probe.astype(np.float64))
    assert why is None and plan[-1] == "hb"                     # float64 signal: one flow, the reference's
    # the host chain itself needs no device when nothing in it does: path and reason are recorded
    ca = da.ConcentrationAnalysis(base, red, lambda x: x, None, None, **{"diff option": "positive"})
    out = ca(da.OpticalImage(probe))
    assert ca.last_path == "generic" and "balancing" in ca.last_path_reason
    assert np.array_equal(out.img, np.maximum(probe - base.img, 0)[..., 0])


def test_estimator_acceptance_rule_in_its_numpy_statement():
    from oracle import phase_correlation_numpy as P

    rng = np.random.default_rng(3)
    a = rng.random((64, 96))
    shifted = np.roll(a, (2, -3), axis=(0, 1))
    (dx, dy), ok, peak = P.phase_correlation(a, shifted, return_peak=True)
    assert ok and abs(dx + 3) < 0.5 and abs(dy - 2) < 0.5 and peak > 0.3
    flat = np.full((64, 96), 0.4)
    _, ok, peak = P.phase_correlation(flat, flat, return_peak=True)
    assert not ok and abs(peak) < 1e-12
    (_, ok, peak) = P.phase_correlation(a, rng.random((64, 96)), return_peak=True)
    n = 64 * 128
    assert not ok and peak < 4 * np.sqrt(np.log(n) / n)
    assert P.phase_correlation(a, rng.random((64, 96)), min_peak_ratio=0.0, max_shift_fraction=1.0)[1] is True
    nan = a.copy()
    nan[3, 4] = np.nan
    assert P.phase_correlation(nan, shifted)[1] is False


def test_rendezvous_file_per_communicator(monkeypatch):
    from darsia_amd import parallel

    monkeypatch.setenv("MASTER_PORT", "29999")
    monkeypatch.delenv("DXH_RENDEZVOUS_FILE", raising=False)
    monkeypatch.setattr(parallel, "_rccl_seq", 0)
    first, second = parallel._rendezvous_path(), parallel._rendezvous_path()
    assert first != second and first.endswith("_0.id") and second.endswith("_1.id")
    monkeypatch.setenv("DXH_RENDEZVOUS_FILE", "/tmp/x.id")
    monkeypatch.setattr(parallel, "_rccl_seq", 0)
    assert parallel._rendezvous_path() == "/tmp/x.id" and parallel._rendezvous_path() == "/tmp/x.id.1"


def test_slab_runs_follow_the_same_policy(monkeypatch):
    """SlabTVD (and bench.py's N > 1 path through it) resolves the dtype flow like split_bregman_tvd: explicit, then
    DXH_PROMOTION, then the reference's own flow wherever the slab path has it."""
    from darsia_amd.parallel import slab_promotion

    monkeypatch.delenv("DXH_PROMOTION", raising=False)
    assert slab_promotion(None, np.float32, 0.1, 1.0, None, 2, 1, False) == "numpy2"
    assert slab_promotion(None, np.float32, np.ones((4, 4)), 1.0, 1.0, 2, 1, False) == "numpy2"      # float64 mu array
    assert slab_promotion(None, np.float64, 0.1, 1.0, None, 2, 1, False) == "native"
    assert slab_promotion(None, np.float32, 0.1, 1.0, None, 2, 3, False) == "native"                  # several sweeps
    assert slab_promotion(None, np.float32, 0.1, 1.0, None, 2, 1, True) == "native"                   # isotropic
    assert slab_promotion(None, np.float32, 0.1, np.ones((4, 4)), 1.0, 2, 1, False) == "native"       # array omega
    assert slab_promotion("native", np.float32, 0.1, 1.0, None, 2, 1, False) == "native"
    with pytest.raises(NotImplementedError):
        slab_promotion("numpy2", np.float32, 0.1, 1.0, None, 2, 1, True)
    monkeypatch.setenv("DXH_PROMOTION", "native")
    assert slab_promotion(None, np.float32, 0.1, 1.0, None, 2, 1, False) == "native"


def test_tvd_operator_copies_and_refresh():
    """ADVICE (round 3): a deep copy of a TVD operator owns its coefficient arrays (an in-place edit of the original's weight
    map must not reach the copy), device contexts and the lock are not shared, the operator survives pickling, and
    refresh() exists for the documented no-in-place-edit contract."""
    import copy
    import pickle

    w = np.full((6, 7), 0.25)
    op = da.TVD(method="heterogeneous bregman", weight=w, omega=np.ones((6, 7)), max_num_iter=3, eps=None)
    cp = copy.deepcopy(op)
    assert cp.weight is not op.weight and np.array_equal(cp.weight, w) and cp.omega is not op.omega
    w[2, 3] = 9.0
    assert cp.weight[2, 3] == 0.25 and op.weight[2, 3] == 9.0
    assert cp._contexts == {} and cp._lock is not op._lock
    back = pickle.loads(pickle.dumps(op))
    assert back.method == op.method and np.array_equal(back.weight, op.weight) and back._contexts == {}
    op.refresh()        # nothing held: a no-op that must not raise
    assert op._contexts == {}
