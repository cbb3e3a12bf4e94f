"""The reference's own dtype flow for float32 images on the device (``promotion="numpy2"``, ``k_fused_np2``): u in
float64, d / b / right-hand side in float32, the shrinkage threshold a Python float or a float64 array
(restoration/split_bregman_tvd.py:84-101,129-146; utils/linear_solvers/jacobi.py:101-114 under NumPy >= 2).

Bar: BIT-EXACT against vectors produced by executing the reference (tests/golden/tvd_golden.npz float32 cases,
tests/golden/tvd_mixed_golden.npz: float32 image x float64 weight map, the workflow's default) -- the same bar the
float64 path has always had.  The native float32 flow is checked against the same vectors at the north-star
tolerance where it holds."""
import os

import numpy as np
import pytest

import darsia_amd as da
from oracle import tvd_numpy as O

pytestmark = pytest.mark.gpu


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL, ATOL = 1e-4, 1e-6


@pytest.fixture(scope="module")
def mixed():
    return np.load(os.path.join(ROOT, "tests", "golden", "tvd_mixed_golden.npz"), allow_pickle=False)


def bits(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape
    if not np.array_equal(a, b):
        d = np.abs(a.astype(np.float64) - b)
        raise AssertionError(f"not bit-exact: {np.count_nonzero(d)} of {d.size} differ, max abs {d.max():.3e}")


SCALAR = [
    ("g2_aniso2d_f32", dict(mu=0.1), 20),
    ("g3_aniso_rgb_f32", dict(mu=0.1), 10),
    ("g3_aniso_rgb_odd_f32", dict(mu=0.15, omega=0.7), 7),
    ("g3_aniso_2ch_odd_f32", dict(mu=0.1), 6),
    ("g2_aniso2d_odd_f32", dict(mu=0.2, ell=0.3), 9),
    ("g2_tiny_f32", dict(mu=0.1), 4),
    ("g2_tiny_col_f32", dict(mu=0.1), 4),
    ("g2_single_px_f32", dict(mu=0.1), 3),
]


@pytest.mark.parametrize("name,kw,n", SCALAR, ids=[c[0] for c in SCALAR])
def test_float32_goldens_of_the_reference_bit_for_bit(golden, name, kw, n):
    img = golden[name + "/img"]
    assert img.dtype == np.float32
    bits(da.split_bregman_tvd(img, max_num_iter=n, promotion="numpy2", **kw), golden[name + "/out"])
    bits(da.split_bregman_tvd(img, max_num_iter=n, **kw), golden[name + "/out"])            # ... which is the DEFAULT call
    # the native flow on the same input: the oracle's same-dtype flow bit for bit, the reference within the tolerance here
    nat = da.split_bregman_tvd(img, max_num_iter=n, promotion="native", **kw)
    bits(nat, O.split_bregman_tvd(img, max_num_iter=n, promotion="strict", **kw))
    np.testing.assert_allclose(nat, golden[name + "/out"], rtol=RTOL, atol=ATOL)


def test_stop_iteration_and_operator_call(golden):
    for name, eps in (("g7_eps_f32", 1e-3), ("g7_eps_rgb_f32", 2e-3)):
        img = golden[name + "/img"]
        ref, _, _, done = O.split_bregman_tvd(img, mu=0.1, max_num_iter=100, eps=eps, return_state=True)
        assert done == int(golden[name + "/iters"])
        bits(da.split_bregman_tvd(img, mu=0.1, max_num_iter=100, eps=eps, promotion="numpy2"), golden[name + "/out"])
    # TVD("heterogeneous bregman") with its regularization default of 1.0 (restoration/tvd.py:37,103-112)
    img = golden["g13_tvd_call/img"]
    op = da.TVD(method="heterogeneous bregman", weight=0.1, max_num_iter=6, eps=None, promotion="numpy2")
    bits(op(img), golden["g13_tvd_call/out"])
    # the 512 x 512 plumbing case of BASELINE configs[0] in float32: crop of the reference's result
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import noisy_blocks  # pure NumPy helper (nothing of the reference is touched at import)
    big32 = noisy_blocks((512, 512), 12, np.float64).astype(np.float32)
    res = da.split_bregman_tvd(big32, mu=0.1, max_num_iter=20, promotion="numpy2")
    bits(res[248:264, 248:264], golden["g12_512/crop_mid_f32"])
    assert float(res.astype(np.float64).sum()) == float(golden["g12_512/sum_f32"])


MIXED = {
    "m1_gray_ell1": dict(ell=1.0, max_num_iter=20),
    "m2_gray_ell02": dict(ell=0.2, omega=0.8, max_num_iter=20),
    "m4_rgb_ell02": dict(ell=0.2, max_num_iter=20),
    "m6_gray_ell1_100": dict(ell=1.0, max_num_iter=100),
    "m8_tvd_operator": dict(ell=1.0, max_num_iter=20),
}


@pytest.mark.parametrize("name", list(MIXED))
def test_float32_image_with_float64_weights_bit_for_bit(mixed, name):
    """The DEFAULT call (no promotion keyword): a float32 image with a float64 mu array takes the reference's flow."""
    img, mu, ref = mixed[name + "/img"], mixed[name + "/mu"], mixed[name + "/out"]
    bits(da.split_bregman_tvd(img, mu=mu, **MIXED[name]), ref)
    # device-resident image in, device-resident result out
    d = da.split_bregman_tvd(da.DeviceArray.from_host(img), mu=mu, **MIXED[name])
    assert isinstance(d, da.DeviceArray) and d.dtype == np.float32
    bits(d.get(), ref)


def test_native_flow_on_the_same_vectors_at_the_north_star_tolerance(mixed):
    for name in ("m1_gray_ell1", "m2_gray_ell02", "m4_rgb_ell02", "m8_tvd_operator"):
        img, mu, ref = mixed[name + "/img"], mixed[name + "/mu"], mixed[name + "/out"]
        got = da.split_bregman_tvd(img, mu=mu, promotion="native", **MIXED[name])
        np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL)
        # ... and it is the oracle's float32 ("strict") flow bit for bit, with the weights rounded to float32 first
        bits(got, O.split_bregman_tvd(img, mu=mu, promotion="strict", **MIXED[name]))
    # ell = None makes ell = 2 mu a float64 ARRAY: float64 enters the right-hand side (k_rhs_np2_het) -- the default call
    # is the reference's result bit for bit; the native flow agrees to 2e-7 (ell = 2 mu is the benign regime)
    for name in ("m3_gray_ellnone", "m5_rgb_ellnone"):
        bits(da.split_bregman_tvd(mixed[name + "/img"], mu=mixed[name + "/mu"], max_num_iter=20), mixed[name + "/out"])
        got = da.split_bregman_tvd(mixed[name + "/img"], mu=mixed[name + "/mu"], max_num_iter=20, promotion="native")
        np.testing.assert_allclose(got, mixed[name + "/out"], rtol=RTOL, atol=ATOL)
        assert np.abs(got - mixed[name + "/out"]).max() < 1e-6


def test_stop_iteration_on_the_mixed_flow(mixed):
    img, mu = mixed["m7_gray_eps/img"], mixed["m7_gray_eps/mu"]
    bits(da.split_bregman_tvd(img, mu=mu, ell=0.2, omega=0.8, max_num_iter=200, eps=2e-3), mixed["m7_gray_eps/out"])
    with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
        st.set_coeffs(mu, 0.8, 0.2, img)
        st.upload(img)
        res = st.run(200, eps=2e-3)
        assert res.converged == 1 and res.iters_done == int(mixed["m7_gray_eps/iters"])


def test_operator_from_build_restoration_on_a_float32_image(mixed):
    """presets/workflows/restoration.py:117-148: porosity weights arrive as a float64 array, the image as float32."""
    from types import SimpleNamespace

    from darsia_amd.presets.workflows import build_restoration
    from darsia_amd.presets.workflows.config import RestorationConfig

    img, mu, ref = mixed["m8_tvd_operator/img"], mixed["m8_tvd_operator/mu"], mixed["m8_tvd_operator/out"]
    rig = SimpleNamespace(baseline=da.ScalarImage(np.zeros(img.shape)), image_porosity=da.ScalarImage(mu),
                          boolean_porosity=da.ScalarImage(mu > 0))
    cfg = RestorationConfig().load({"method": "tvd", "options": {"weight": "image_porosity", "max_num_iter": 20,
                                                                 "eps": 1e-30, "omega": 1.0, "regularization": 1.0}})
    op = build_restoration(cfg, rig)
    assert op.method == "heterogeneous bregman" and op.weight.dtype == np.float64
    bits(op(img), ref)
    out_img = op(da.ScalarImage(img))
    bits(out_img.img, ref)


def test_float32_mu_array_and_windows_at_100_iterations():
    """(a) an explicit numpy2 call with a float32 mu array: NumPy forms mu / ell in float32 before it meets the float64
    iterate; (b) 512 x 512 windows of a larger image after 100 iterations at ell = 1.0 -- the regime where the native flow
    leaves the tolerance -- bit for bit against the oracle's reference flow (dependence cone: tests/test_tvd_gpu.py)."""
    rng = np.random.default_rng(17)
    img = rng.random((90, 130)).astype(np.float32)
    mu32 = (0.02 + 0.1 * rng.random(img.shape)).astype(np.float32)
    bits(da.split_bregman_tvd(img, mu=mu32, ell=0.7, omega=1.2, max_num_iter=15, promotion="numpy2"),
         O.split_bregman_tvd(img, mu=mu32, ell=0.7, omega=1.2, max_num_iter=15))
    from darsia_amd.synthetic import synthetic_image

    H, W, n, m, S = 1500, 2100, 100, 102, 512
    big = synthetic_image(H, W, 1)
    por = 0.3 + 0.7 * np.random.default_rng(4).random((H, W))
    por[400:700, 300:900] = 0.0
    mu = 0.1 * por
    out = da.split_bregman_tvd(big, mu=mu, ell=1.0, max_num_iter=n)
    assert out.dtype == np.float32
    worst_native = 0.0
    native = da.split_bregman_tvd(big, mu=mu, ell=1.0, max_num_iter=n, promotion="native")
    for (r0, c0) in ((0, 0), (300, 500), (H - S, W - S)):
        ra, rb = (0 if r0 == 0 else m), (S if r0 + S == H else S - m)
        ca, cb = (0 if c0 == 0 else m), (S if c0 + S == W else S - m)
        want = O.split_bregman_tvd(np.ascontiguousarray(big[r0:r0 + S, c0:c0 + S]), mu=np.ascontiguousarray(mu[r0:r0 + S, c0:c0 + S]),
                                   ell=1.0, max_num_iter=n)[ra:rb, ca:cb]
        bits(out[r0 + ra:r0 + rb, c0 + ca:c0 + cb], want)
        nat = native[r0 + ra:r0 + rb, c0 + ca:c0 + cb]
        worst_native = max(worst_native, float((np.abs(nat.astype(np.float64) - want) / (ATOL + RTOL * np.abs(want))).max()))
    print(f"native float32 flow after {n} iterations at ell = 1.0: worst error / tolerance = {worst_native:.2f}")


def test_negative_thresholds_take_the_general_shrinkage():
    """A float64 weight map with negative entries (np.maximum(|x| - k, 0) with k < 0 GROWS x) and a -0.0: the K-stage
    pipeline's clamp form of the shrinkage needs k >= 0, rows that hold anything else fall back to the expression as written
    (wave-uniform per row, PipeNp2::step) -- bit for bit like the oracle.  (A NaN weight is outside what is reproduced: the
    compressed state re-derives d = shrink(0, NaN) = NaN at the first iteration where the reference still holds d = 0, so
    the NaN reaches the iterate one iteration early -- DESIGN.md, known deviations.)"""
    rng = np.random.default_rng(23)
    for shape, n in (((70, 150), 9), ((40, 90, 3), 9), ((300, 700, 3), 13)):
        img = rng.random(shape).astype(np.float32)
        mu = 0.1 * rng.random(shape)
        mu[10:14, 20:60] *= -1.0
        mu[30, 5] = -0.0
        got = da.split_bregman_tvd(img, mu=mu, ell=0.5, max_num_iter=n)
        bits(got, O.split_bregman_tvd(img, mu=mu, ell=0.5, max_num_iter=n))


def test_two_and_four_channel_images_in_the_k_stage_pipeline():
    """2- and 4-channel float32 images (scalar coefficients): K = 4 launches of the numpy2 pipeline, bit for bit like the
    oracle, and launch counts that show the pipeline served them (round 3: one iteration per launch)."""
    rng = np.random.default_rng(31)
    for shape, n in (((120, 333, 2), 11), ((90, 257, 4), 14), ((700, 1500, 4), 8)):
        img = rng.random(shape).astype(np.float32)
        with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
            st.set_coeffs(0.15, 0.8, 0.4, img)
            st.upload(img)
            res = st.run(n)
            assert res.launches == n // 4 + n % 4
            if shape[0] < 200:
                bits(st.download(), O.split_bregman_tvd(img, mu=0.15, omega=0.8, ell=0.4, max_num_iter=n))
            else:
                S = 160
                got = st.download()
                bits(got[:S - n - 2, :S - n - 2], O.split_bregman_tvd(np.ascontiguousarray(img[:S, :S]), mu=0.15, omega=0.8, ell=0.4,
                                                                      max_num_iter=n)[:S - n - 2, :S - n - 2])
        # coupled sums off (an image whose channels the caller treats as separate scalar fields is not a thing in the
        # reference: the K-stage launch equals the one-iteration-per-launch kernel either way)
    img = rng.random((64, 100, 2)).astype(np.float32)
    a = da.split_bregman_tvd(img, mu=0.2, max_num_iter=8)
    import os as _os
    _os.environ["DXH_FUSE_K"] = "1"
    try:
        b = da.split_bregman_tvd(img, mu=0.2, max_num_iter=8)
    finally:
        del _os.environ["DXH_FUSE_K"]
    bits(a, b)


def test_device_resident_coefficient_arrays_and_warm_starts(mixed):
    """Device-resident chaining in the default flow: a float64 weight map that lives in HBM (``DeviceArray``), a float32 one
    (the threshold becomes the float32 quotient, formed on the device), and a warm start whose u0 / d0 / b0 stay on the
    device from call to call -- same bits as the host-array calls, which are the reference's."""
    img, mu, ref = mixed["m1_gray_ell1/img"], mixed["m1_gray_ell1/mu"], mixed["m1_gray_ell1/out"]
    d_img, d_mu = da.DeviceArray.from_host(img), da.DeviceArray.from_host(mu)
    out = da.split_bregman_tvd(d_img, mu=d_mu, ell=1.0, max_num_iter=20)
    assert isinstance(out, da.DeviceArray)
    bits(out.get(), ref)
    bits(da.split_bregman_tvd(img, mu=d_mu, ell=1.0, max_num_iter=20), ref)          # host image, device weights
    mu32 = mu.astype(np.float32)
    want32 = O.split_bregman_tvd(img, mu=mu32, ell=0.7, max_num_iter=15)
    bits(da.split_bregman_tvd(img, mu=mu32, ell=0.7, max_num_iter=15, promotion="numpy2"), want32)
    bits(da.split_bregman_tvd(d_img, mu=da.DeviceArray.from_host(mu32), ell=0.7, max_num_iter=15, promotion="numpy2").get(), want32)
    # ell = None next to a float64 device map: ell = 2 mu is an array (k_rhs_np2_het)
    img3, mu3, ref3 = mixed["m3_gray_ellnone/img"], mixed["m3_gray_ellnone/mu"], mixed["m3_gray_ellnone/out"]
    bits(da.split_bregman_tvd(img3, mu=da.DeviceArray.from_host(mu3), max_num_iter=20), ref3)
    # warm-started chain with the state on the device: 5 + 5 iterations == the host chain == 10 iterations in one call
    z = lambda: da.DeviceArray.from_host(np.zeros(img.shape + (2,), np.float32))   # noqa: E731
    d0, b0 = z(), z()
    first = da.split_bregman_tvd(d_img, mu=0.1, max_num_iter=5, x0=(d_img, d0, b0))
    assert isinstance(first, da.DeviceArray)
    second = da.split_bregman_tvd(d_img, mu=0.1, max_num_iter=5, x0=(first, d0, b0))
    hd, hb = np.zeros(img.shape + (2,), np.float32), np.zeros(img.shape + (2,), np.float32)
    h1 = da.split_bregman_tvd(img, mu=0.1, max_num_iter=5, x0=(img.copy(), hd, hb))
    h2 = da.split_bregman_tvd(img, mu=0.1, max_num_iter=5, x0=(h1, hd, hb))
    bits(first.get(), h1)
    bits(second.get(), h2)
    bits(d0.get(), hd)
    bits(b0.get(), hb)
    # ... and the host chain is the oracle's chain (planar split variables there)
    zp = [np.zeros(img.shape, np.float32) for _ in range(2)]
    o1, od, ob, _ = O.split_bregman_tvd(img, mu=0.1, max_num_iter=5, x0=(img.copy(), zp, [a.copy() for a in zp]), return_state=True)
    o2 = O.split_bregman_tvd(img, mu=0.1, max_num_iter=5, x0=(o1, od, ob))
    bits(h1, o1)
    bits(h2, o2)


def test_unsupported_configurations_say_so():
    img = np.random.default_rng(1).random((20, 30)).astype(np.float32)
    with pytest.raises(NotImplementedError, match="numpy2"):
        da.split_bregman_tvd(img, mu=0.1, isotropic=True, promotion="numpy2")
    with pytest.raises(NotImplementedError, match="numpy2"):
        da.split_bregman_tvd(img, mu=0.1, solver=da.MG(2), promotion="numpy2")
    with pytest.raises(NotImplementedError, match="numpy2"):
        da.split_bregman_tvd(img.astype(np.float64), mu=0.1, promotion="numpy2")
    with pytest.raises(ValueError):
        da.split_bregman_tvd(img, mu=0.1, promotion="float16")


def test_small_random_sweep_of_the_numpy2_flow_and_the_volume_kernels():
    """A short run of tests/fuzz/fuzz_r3.py inside the suite (80 cases: the numpy2 flow on random 2-D / colour float32 images
    with scalar, float32-array and float64-array mu, special magnitudes, eps; dim = 3 volumes through the vectorised and
    the scalar march / sweeps) -- bit for bit against the oracle."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_r3.py"), "7", "80"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "80 cases, 0 mismatches" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_operator_keeps_its_numpy2_context_between_calls(mixed):
    """A TVD operator object is long-lived in the reference's workflows (presets/workflows/restoration.py returns one per run);
    in the numpy2 flow it keeps its device context -- the float64 weight map's planes -- from frame to frame.  Same bits as
    fresh calls; a new weight array, another shape or release() start afresh."""
    img, mu, ref = mixed["m8_tvd_operator/img"], mixed["m8_tvd_operator/mu"], mixed["m8_tvd_operator/out"]
    op = da.TVD(method="heterogeneous bregman", weight=mu, max_num_iter=20, eps=None)
    bits(op(img), ref)
    held = op._contexts["numpy2"][2]
    img2 = np.ascontiguousarray(img[::-1])
    want2 = da.split_bregman_tvd(img2, mu=mu, ell=1.0, max_num_iter=20)
    bits(op(img2), want2)
    assert op._contexts["numpy2"][2] is held                       # the same context served the second frame
    bits(op(img), ref)                                             # ... and leaves no trace of it
    op.weight = mu * 0.5                                           # another weight map: coefficients are set again
    bits(op(img), da.split_bregman_tvd(img, mu=mu * 0.5, ell=1.0, max_num_iter=20))
    small = np.ascontiguousarray(img[:40, :50])
    op.weight = np.ascontiguousarray(mu[:40, :50])
    bits(op(small), da.split_bregman_tvd(small, mu=op.weight, ell=1.0, max_num_iter=20))
    assert op._contexts["numpy2"][2] is not held                   # another shape: another context, the old one closed
    import copy
    assert copy.deepcopy(op)._contexts == {}
    op.release()
    assert op._contexts == {}


@pytest.mark.parametrize("shape,n,nbands", [((300, 200, 3), 7, 3), ((257, 129), 5, 4), ((96, 64, 2), 6, 2), ((40, 33, 3), 9, 0)],
                         ids=["rgb", "gray-odd", "2ch", "auto"])
def test_row_bands_of_the_numpy2_flow_equal_the_one_piece_run_and_the_oracle(shape, n, nbands):
    """dxh_tvd_solve_banded_typed: float32 host rows through float64 band contexts (promotion numpy2), each band with the
    dependence cone of n iterations as halo -- the whole-image result bit for bit."""
    from darsia_amd.restoration.split_bregman_tvd import solve_banded_numpy2

    rng = np.random.default_rng(17)
    img = rng.random(shape, dtype=np.float32)
    for kw in (dict(mu=0.1, omega=1.0, ell=None), dict(mu=1.0, omega=0.7, ell=0.45)):
        out, used = solve_banded_numpy2(img, kw["mu"], kw["omega"], kw["ell"], n, nbands=nbands, return_bands=True)
        assert out.dtype == np.float32 and (nbands == 0 or used == nbands)
        bits(out, O.split_bregman_tvd(img, max_num_iter=n, promotion="numpy2", **kw))
        with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
            st.set_coeffs(kw["mu"], kw["omega"], kw["ell"], img)
            st.upload(img)
            st.run(n)
            bits(out, st.download())


def test_large_host_image_takes_the_row_bands_by_default():
    """>= 32 MB float32 host image, scalar coefficients, fixed iteration count: the default call goes through the bands
    (a size the oracle cannot check in seconds: against the one-piece device run, bit for bit)."""
    rng = np.random.default_rng(3)
    img = rng.random((1500, 2000, 3), dtype=np.float32)
    assert img.nbytes >= (32 << 20)
    out = da.split_bregman_tvd(img, mu=0.1, max_num_iter=12)
    with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
        st.set_coeffs(0.1, 1.0, None, img)
        st.upload(img)
        st.run(12)
        bits(out, st.download())


# ---- BASELINE.json full size (4000 x 8000 x 3 float32) in the flow a caller of the drop-in API gets -------------------------
def test_full_size_config2_default_call_windows_against_the_oracle(monkeypatch):
    """configs[1] at its real size through the DEFAULT call ``da.split_bregman_tvd(img32, mu=0.1, max_num_iter=100)`` -- the
    numpy2 flow, row bands for a host image of this size: (a) four 512 x 512 windows (three corners and the centre) BIT FOR BIT
    against the oracle's reference flow on the crop (dependence cone, restoration/split_bregman_tvd.py:84-101,129-146,
    utils/linear_solvers/jacobi.py:41-54,101-114: after n iterations a pixel knows nothing beyond n + 1 pixels;
    tests/test_tvd_gpu.py states the argument); (b) the one-context K = 4 pipeline run == the one-iteration-per-launch run ==
    the banded default call, bit for bit over the whole image; (c) 3 iterations over the WHOLE image against the oracle."""
    from darsia_amd.synthetic import synthetic_image

    H, W, n, m, S = 4000, 8000, 100, 102, 512
    img = synthetic_image(H, W, 3)
    out = da.split_bregman_tvd(img, mu=0.1, max_num_iter=n)          # the default call (banded at this size)
    assert out.dtype == np.float32 and np.isfinite(out).all()

    def one_context(iters):
        with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
            st.set_coeffs(0.1, 1.0, None, img)
            st.upload(img)
            res = st.run(iters)
            return st.download(), res

    out_k4, res = one_context(n)
    assert res.iters_done == n and res.launches == n // 4               # the K = 4 NP2 pipeline served every launch
    bits(out, out_k4)
    monkeypatch.setenv("DXH_FUSE_K", "1")
    out_k1, res1 = one_context(n)
    monkeypatch.delenv("DXH_FUSE_K")
    assert res1.launches == n
    bits(out_k4, out_k1)
    del out_k1, out_k4
    for (r0, c0) in ((0, 0), ((H - S) // 2, (W - S) // 2), (H - S, W - S), (0, W - S)):
        crop = np.ascontiguousarray(img[r0:r0 + S, c0:c0 + S])
        ra, rb = (0 if r0 == 0 else m), (S if r0 + S == H else S - m)
        ca, cb = (0 if c0 == 0 else m), (S if c0 + S == W else S - m)
        want = O.split_bregman_tvd(crop, mu=0.1, max_num_iter=n)[ra:rb, ca:cb]      # the reference's NumPy-2 dtype flow
        bits(out[r0 + ra:r0 + rb, c0 + ca:c0 + cb], want)
    out3, _ = one_context(3)
    bits(out3, O.split_bregman_tvd(img, mu=0.1, max_num_iter=3))


def test_full_size_gray_and_float64_images_windows_against_the_oracle():
    """The two other instantiations of the LDS-row pipeline at BASELINE's width: (a) a 4000 x 8000 single-channel float32
    image in the default flow, 100 iterations = 16 launches of six + one of four (kNp2GrayK; the hand-off rows forwarded
    and single-buffered); (b) a 1500 x 8000 x 3 float64 image, 40 iterations (every quantity float64, `F = double`).  Oracle
    windows as in the RGB test above; launch counts show which pipeline served."""
    from darsia_amd.synthetic import synthetic_image

    H, W, n, S = 4000, 8000, 100, 384
    img = synthetic_image(H, W, 1)
    with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
        st.set_coeffs(0.1, 1.0, None, img)
        st.upload(img)
        res = st.run(n)
        out = st.download()
    assert res.launches == 17 and np.isfinite(out).all()
    m = n + 2
    for (r0, c0) in ((0, 0), ((H - S) // 2, (W - S) // 2), (H - S, W - S)):
        crop = np.ascontiguousarray(img[r0:r0 + S, c0:c0 + S])
        ra, rb = (0 if r0 == 0 else m), (S if r0 + S == H else S - m)
        ca, cb = (0 if c0 == 0 else m), (S if c0 + S == W else S - m)
        bits(out[r0 + ra:r0 + rb, c0 + ca:c0 + cb], O.split_bregman_tvd(crop, mu=0.1, max_num_iter=n)[ra:rb, ca:cb])
    H, n, S = 1500, 40, 256
    img64 = synthetic_image(H, W, 3).astype(np.float64)
    with da.DeviceTVD(img64.shape, img64.dtype) as st:
        st.set_coeffs(0.1, 1.0, None, img64)
        st.upload(img64)
        res = st.run(n)
        out = st.download()
    assert res.launches == 10
    m = n + 2
    for (r0, c0) in ((0, 0), ((H - S) // 2, (W - S) // 2), (H - S, W - S), (H - S, 0)):
        crop = np.ascontiguousarray(img64[r0:r0 + S, c0:c0 + S])
        ra, rb = (0 if r0 == 0 else m), (S if r0 + S == H else S - m)
        ca, cb = (0 if c0 == 0 else m), (S if c0 + S == W else S - m)
        bits(out[r0 + ra:r0 + rb, c0 + ca:c0 + cb], O.split_bregman_tvd(crop, mu=0.1, max_num_iter=n)[ra:rb, ca:cb])


def test_large_host_image_with_array_valued_omega_or_ell():
    """>= 32 MB float32 host image, scalar mu next to a float64 ARRAY omega (or ell): not a row-band case (the bands take
    scalar coefficients) -- the kept-context path serves it, same bits as the explicit device run.  (Regression: the band
    shortcut once looked at mu only and failed on ``float(omega)``.)"""
    rng = np.random.default_rng(5)
    img = rng.random((1500, 2000, 3), dtype=np.float32)
    assert img.nbytes >= (32 << 20)
    om = 0.5 + rng.random(img.shape)
    out = da.split_bregman_tvd(img, mu=0.1, omega=om, ell=0.3, max_num_iter=3)
    S = 96
    bits(out[:S, :S], O.split_bregman_tvd(np.ascontiguousarray(img[:S + 8, :S + 8]), mu=0.1, omega=np.ascontiguousarray(om[:S + 8, :S + 8]),
                                          ell=0.3, max_num_iter=3)[:S, :S])
    ell = 0.2 + 0.5 * rng.random(img.shape)
    out = da.split_bregman_tvd(img, mu=0.1, ell=ell, max_num_iter=3)
    bits(out[-S:, -S:], O.split_bregman_tvd(np.ascontiguousarray(img[-S - 8:, -S - 8:]), mu=0.1, ell=np.ascontiguousarray(ell[-S - 8:, -S - 8:]),
                                            max_num_iter=3)[-S:, -S:])


# ---- float32 images outside the default solve: several Jacobi sweeps, volumes, six channels (three kinds of launches) ----
from test_oracle_np2_generic_golden import (ARRAY_CASES, CASES as GENERIC_CASES, H1_CASES, JACOBI_CASES, TOL_CASES,  # noqa: E402
                                            WARM_CASES, ADAPT_CASES, adapt_inputs, array_kwargs, case_kwargs, warm_inputs)


@pytest.fixture(scope="module")
def generic():
    return np.load(os.path.join(ROOT, "tests", "golden", "tvd_np2_generic_golden.npz"), allow_pickle=False)


@pytest.mark.parametrize("name", sorted(GENERIC_CASES))
def test_several_sweeps_volumes_and_six_channels_bit_for_bit_against_the_reference(generic, name):
    """k_rhs_np2 / k_jacobi_np2_first + the float64 sweeps / k_shrink_np2: vectors produced by executing the reference
    (tests/golden/make_golden_np2_generic.py).  The DEFAULT call takes this flow."""
    kw, m = case_kwargs(generic, name)
    img = generic[name + "/img"]
    out = da.split_bregman_tvd(img, solver=da.Jacobi(maxiter=m), **kw)
    bits(out, generic[name + "/out"])
    bits(da.split_bregman_tvd(img, solver=da.Jacobi(maxiter=m), promotion="numpy2", **kw), generic[name + "/out"])
    # device in, device out
    dev = da.split_bregman_tvd(da.DeviceArray.from_host(img), solver=da.Jacobi(maxiter=m), **kw)
    bits(dev.get(), generic[name + "/out"])


def test_stop_iteration_with_several_sweeps(generic):
    img = generic["n12_gray_m3_eps/img"]
    out = da.split_bregman_tvd(img, mu=0.1, max_num_iter=200, eps=2e-3, solver=da.Jacobi(maxiter=3))
    bits(out, generic["n12_gray_m3_eps/out"])
    with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
        st.set_coeffs(0.1, 1.0, None, img)
        st.upload(img)
        r = st.run(200, 3, None, eps=2e-3)
        assert r.converged and r.iters_done == int(generic["n12_gray_m3_eps/iters"])


@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_of_the_generic_numpy2_flow_against_the_oracle(seed):
    rng = np.random.default_rng(100 + seed)
    for _ in range(6):
        dim = int(rng.choice([2, 3]))
        if dim == 3:
            shape = tuple(int(v) for v in rng.integers(3, 40, size=3))
        else:
            shape = (int(rng.integers(2, 70)), int(rng.integers(2, 90))) + ((int(rng.choice([1, 2, 3, 4, 5, 7])),) if rng.random() < 0.6 else ())
        img = rng.random(shape, dtype=np.float32) * np.float32(rng.choice([1.0, 255.0, 1e-3]))
        m = int(rng.choice([1, 2, 3, 4, 5, 6, 9]))
        mu = float(rng.choice([0.05, 0.3, 1.0])) * float(img.max())
        kw = dict(mu=mu, omega=float(rng.choice([1.0, 0.6])), ell=None if rng.random() < 0.5 else float(rng.choice([0.4, 1.0])) * float(img.max()))
        if kw["ell"] is not None and rng.random() < 0.4:
            kw["mu"] = mu * (0.2 + 0.8 * rng.random(shape))           # float64 array
        n = int(rng.integers(1, 7))
        out = da.split_bregman_tvd(img, dim=dim, max_num_iter=n, solver=da.Jacobi(maxiter=m), **kw)
        ref = O.split_bregman_tvd(img, dim=dim, max_num_iter=n, jacobi_maxiter=m, promotion="numpy2", **kw)
        bits(out, ref)


def test_verbose_run_is_the_same_flow_and_prints_the_reference_messages(mixed, generic, capsys):
    """verbose=True (the reference's debug path: one iteration per call, increment and energy printed) stays in the
    reference's dtype flow: the vectors m7 / n12 were recorded from verbose runs of the reference with eps."""
    img, mu = mixed["m7_gray_eps/img"], mixed["m7_gray_eps/mu"]
    out = da.split_bregman_tvd(img, mu=mu, ell=0.2, omega=0.8, max_num_iter=200, eps=2e-3, verbose=True)
    printed = capsys.readouterr().out
    bits(out, mixed["m7_gray_eps/out"])
    lines = [ln for ln in printed.splitlines() if ln.startswith("Split Bregman iteration")]
    assert len(lines) == int(mixed["m7_gray_eps/iters"]) and printed.startswith("The energy functional starts at ")
    img = generic["n12_gray_m3_eps/img"]
    out = da.split_bregman_tvd(img, mu=0.1, max_num_iter=200, eps=2e-3, solver=da.Jacobi(maxiter=3), verbose=5)
    printed = capsys.readouterr().out
    bits(out, generic["n12_gray_m3_eps/out"])
    assert sum(ln.startswith("Split Bregman iteration") for ln in printed.splitlines()) == (int(generic["n12_gray_m3_eps/iters"]) + 4) // 5


def test_the_round_one_float32_vector_with_three_sweeps_is_bit_exact_now(golden):
    """g6_jacobi3_rgb_f32 (tests/golden/make_golden.py: RGB float32, Jacobi(maxiter=3)) was within the tolerance through the
    native kernels since round 1; through the default flow it is the reference's result bit for bit."""
    img = golden["g6_jacobi3_rgb_f32/img"]
    bits(da.split_bregman_tvd(img, mu=0.1, max_num_iter=5, solver=da.Jacobi(maxiter=3)), golden["g6_jacobi3_rgb_f32/out"])


@pytest.mark.parametrize("name", sorted(TOL_CASES))
def test_tolerance_in_the_jacobi_solver_bit_for_bit_against_the_reference(generic, name):
    c = TOL_CASES[name]
    out = da.split_bregman_tvd(generic[name + "/img"], solver=da.Jacobi(maxiter=c["m"], tol=c["tol"]), **c["kw"])
    bits(out, generic[name + "/out"])


@pytest.mark.parametrize("name", sorted(JACOBI_CASES))
def test_jacobi_on_float32_operands_returns_the_references_float64_iterate(generic, name):
    """Jacobi.__call__ (jacobi.py:74-133) on float32 arrays under NumPy >= 2: float64 out -- dxh_jacobi_solve_np2."""
    c = JACOBI_CASES[name]
    solver = da.Jacobi(maxiter=c["m"], tol=c["tol"])
    solver.update_params(mass_coeff=c["mass"], diffusion_coeff=c["diff"], dim=c["dim"])
    x0, rhs = generic[name + "/x0"], generic[name + "/rhs"]
    out = solver(x0, rhs)
    assert out.dtype == np.float64
    bits(out, generic[name + "/out"])
    dev = solver(da.DeviceArray.from_host(x0), da.DeviceArray.from_host(rhs))
    assert dev.dtype == np.float64
    bits(dev.get(), generic[name + "/out"])
    # the float32 kernels on request: the oracle's same-dtype flow
    nat = solver(x0, rhs, promotion="native")
    assert nat.dtype == np.float32
    bits(nat, O.jacobi(x0, rhs, c["mass"], c["diff"], dim=c["dim"], maxiter=c["m"], tol=c["tol"], promotion="strict").astype(np.float32))


@pytest.mark.parametrize("name", sorted(H1_CASES))
def test_h1_regularization_of_float32_images_bit_for_bit_against_the_reference(generic, name):
    c = H1_CASES[name]
    img = generic[name + "/img"]
    out = da.H1_regularization(img, solver=da.Jacobi(maxiter=c["m"], tol=c["tol"]), **c["kw"])
    assert out.dtype == np.float32
    bits(out, generic[name + "/out"])
    if c["tol"] is None or img.ndim == c["kw"]["dim"]:
        dev = da.H1_regularization(da.DeviceArray.from_host(img), solver=da.Jacobi(maxiter=c["m"], tol=c["tol"]), **c["kw"])
        bits(dev.get(), generic[name + "/out"])


@pytest.mark.parametrize("name", sorted(ARRAY_CASES))
def test_array_valued_omega_and_ell_bit_for_bit_against_the_reference(generic, name):
    """float64 arrays (a*): parts of the right-hand side turn float64 (k_rhs_np2_het, the double context's coefficient
    planes); float32 arrays only (f*): the reference stays float32 throughout and the DEFAULT call takes the native kernels,
    which are that flow."""
    kw, m = array_kwargs(generic, name)
    img = generic[name + "/img"]
    bits(da.split_bregman_tvd(img, solver=da.Jacobi(maxiter=m), **kw), generic[name + "/out"])
    bits(da.split_bregman_tvd(img, solver=da.Jacobi(maxiter=m), promotion="numpy2", **kw), generic[name + "/out"])
    bits(da.split_bregman_tvd(da.DeviceArray.from_host(img), solver=da.Jacobi(maxiter=m), **kw).get(), generic[name + "/out"])


def test_mixed_float32_and_float64_coefficient_arrays_take_the_native_flow_and_say_so(caplog):
    rng = np.random.default_rng(5)
    img = rng.random((24, 30), dtype=np.float32)
    mu, ell = 0.05 + 0.1 * rng.random((24, 30)), (0.3 + 0.3 * rng.random((24, 30))).astype(np.float32)
    with pytest.raises(NotImplementedError, match="numpy2"):
        da.split_bregman_tvd(img, mu=mu, ell=ell, max_num_iter=3, promotion="numpy2")
    import logging
    with caplog.at_level(logging.INFO):
        out = da.split_bregman_tvd(img, mu=mu, ell=ell, max_num_iter=3)
    assert "float64 arrays only" in caplog.text
    np.testing.assert_allclose(out, O.split_bregman_tvd(img, mu=mu, ell=ell, max_num_iter=3, promotion="numpy2"), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("name", sorted(WARM_CASES))
def test_warm_start_bit_for_bit_against_the_reference(generic, name):
    """x0 = (u0, d0, b0) with the float32 d0 / b0 a previous call left behind: the result AND what the call leaves in the
    caller's d0 / b0 (the reference updates them in place, split_bregman_tvd.py:92-96,143-146)."""
    kw, m, n = warm_inputs(generic, name)
    d0, b0 = generic[name + "/d0"].copy(), generic[name + "/b0"].copy()
    out = da.split_bregman_tvd(generic[name + "/img"], x0=(generic[name + "/u0"], d0, b0), solver=da.Jacobi(maxiter=m),
                               max_num_iter=n, **kw)
    bits(out, generic[name + "/out"])
    bits(d0, generic[name + "/d"])
    bits(b0, generic[name + "/b"])


def test_two_warm_started_halves_equal_the_reference_chain_and_zero_iterations_change_nothing(generic):
    img = generic["w1_gray/img"]
    d0, b0 = np.zeros(img.shape + (2,), np.float32), np.zeros(img.shape + (2,), np.float32)
    u1 = da.split_bregman_tvd(img, mu=0.1, x0=(img.copy(), d0, b0), max_num_iter=4)
    bits(u1, generic["w1_gray/u0"])
    bits(d0, generic["w1_gray/d0"])
    bits(b0, generic["w1_gray/b0"])
    d1, b1 = d0.copy(), b0.copy()
    u2 = da.split_bregman_tvd(img, mu=0.1, x0=(u1, d1, b1), max_num_iter=0)
    bits(u2, u1)
    bits(d1, d0)
    bits(b1, b0)


@pytest.mark.parametrize("name", sorted(ADAPT_CASES))
def test_adaptive_regularisation_bit_for_bit_against_the_reference(generic, name):
    """adaptive ell on a float32 image (split_bregman_tvd.py:149-158): ell a float32 array, explicit float32 d / b planes on
    the device (k_rhs_np2<.., FROM_DB>, k_shrink_db_np2, k_adapt_ell_np2)."""
    kw, m, sched = adapt_inputs(generic, name)
    bits(da.split_bregman_tvd(generic[name + "/img"], solver=da.Jacobi(maxiter=m), adaptive=sched, **kw), generic[name + "/out"])


def test_the_round_one_adaptive_float32_vectors_are_bit_exact_now(golden):
    """g14_adaptive_f32 / g14_adaptive_rgb_f32: ell = 1 / |grad u| makes the iteration expansive (values reach 1e7), the
    native float32 flow decorrelates from the reference in a few elements -- the default flow reproduces it bit for bit."""
    for name, kw, sched in (("g14_adaptive_f32", dict(mu=0.1, max_num_iter=9), lambda it: it % 3 == 1),
                            ("g14_adaptive_rgb_f32", dict(mu=0.1, omega=0.9, max_num_iter=6), lambda it: it >= 2)):
        bits(da.split_bregman_tvd(golden[name + "/img"], adaptive=sched, **kw), golden[name + "/out"])


def test_a_tolerance_met_by_the_first_sweep_hands_the_float32_iterate_back():
    """jacobi.py:123-126: the reference returns x0 ITSELF then -- a float32 array -- and the TV loop's D u + b, shrinkage
    included, runs in float32 for that iteration (the stored q is marked float32-formed: Np2Coef::q32).  Found by the fuzzer."""
    rng = np.random.default_rng(12)
    img = rng.random((37, 53, 3), dtype=np.float32)
    for kw in (dict(mu=0.1, ell=0.3), dict(mu=0.1, ell=0.2, omega=0.7), dict(mu=0.1 + 0.1 * rng.random(img.shape), ell=0.3)):
        for n in (1, 2, 4):
            ref = O.split_bregman_tvd(img, max_num_iter=n, jacobi_maxiter=3, jacobi_tol=0.5, promotion="numpy2", **kw)
            bits(da.split_bregman_tvd(img, max_num_iter=n, solver=da.Jacobi(maxiter=3, tol=0.5), **kw), ref)
    # warm-started chain through that state, d / b included
    kw = dict(mu=0.1, ell=0.3)
    _, dl, bl, _ = O.split_bregman_tvd(img, max_num_iter=2, jacobi_maxiter=3, jacobi_tol=0.5, promotion="numpy2", return_state=True, **kw)
    u1 = O.split_bregman_tvd(img, max_num_iter=2, jacobi_maxiter=3, jacobi_tol=0.5, promotion="numpy2", **kw)
    ref2, dl2, bl2, _ = O.split_bregman_tvd(img, max_num_iter=1, jacobi_maxiter=3, jacobi_tol=0.5, promotion="numpy2", return_state=True,
                                            x0=(u1, [a.copy() for a in dl], [a.copy() for a in bl]), **kw)
    d0, b0 = np.stack(dl, axis=-1), np.stack(bl, axis=-1)
    out2 = da.split_bregman_tvd(img, max_num_iter=1, solver=da.Jacobi(maxiter=3, tol=0.5), x0=(u1, d0, b0), **kw)
    bits(out2, ref2)
    bits(d0, np.stack(dl2, axis=-1))
    bits(b0, np.stack(bl2, axis=-1))
    # the solver alone
    sol = da.Jacobi(maxiter=5, tol=10.0)
    sol.update_params(mass_coeff=1.0, diffusion_coeff=0.3, dim=2)
    x0 = img[..., 0].copy()
    out = sol(x0, x0)
    assert out.dtype == np.float32 and np.array_equal(out, x0)


# ---- Jacobi(maxiter = 2 .. 6) on 1 / 3 channels with scalar coefficients: ONE pass per iteration (dxh_np2_sweeps.h) ----
@pytest.mark.parametrize("sweeps", [2, 3, 4, 5, 6])
@pytest.mark.parametrize("shape", [(41, 700, 3), (300, 517), (23, 60, 3), (2, 2), (9, 131, 1)])
@pytest.mark.parametrize("mu_kind", ["scalar", "array", "array with negative entries"])
def test_several_sweeps_run_as_one_pass_per_iteration_bit_for_bit_against_the_oracle(sweeps, shape, mu_kind, monkeypatch):
    """Strips of both kinds (image-border and interior waves), chunk starts inside the image (DXH_NCHUNKS), the first
    iteration on the float32 image and later calls on the float64 iterate, the reference's replicate rules at every border --
    against the oracle's reference flow (jacobi.py:107-114 inside split_bregman_tvd.py:129-146); the launch count shows the
    pipeline served the call, and the three-launch flow it replaces (DXH_FUSED_TUNE bit 33554432) gives the same bits."""
    rng = np.random.default_rng(sweeps * 100 + shape[1])
    img = rng.random(shape, dtype=np.float32)
    kw = dict(mu=0.13, omega=0.9, ell=0.35)
    if mu_kind != "scalar":   # a float64 weight map next to scalar omega / ell: the threshold mu / ell is a plane
        kw["mu"] = 0.13 * (0.2 + 0.8 * rng.random(shape))
        if mu_kind.endswith("negative entries"):
            kw["mu"][rng.random(shape) < 0.02] *= -1.0
            kw["mu"].flat[0] = -0.0
    n = 4
    ref, dl, bl, _ = O.split_bregman_tvd(img, max_num_iter=n, jacobi_maxiter=sweeps, promotion="numpy2", return_state=True, **kw)
    ref2 = O.split_bregman_tvd(img, max_num_iter=n + 3, jacobi_maxiter=sweeps, promotion="numpy2", **kw)
    for chunks, tune in ((None, None), ("2", None), (None, "33554432")):
        if chunks and shape[0] < 200:
            continue
        monkeypatch.delenv("DXH_NCHUNKS", raising=False)
        monkeypatch.delenv("DXH_FUSED_TUNE", raising=False)
        if chunks:
            monkeypatch.setenv("DXH_NCHUNKS", chunks)
        if tune:
            monkeypatch.setenv("DXH_FUSED_TUNE", tune)
        with da.DeviceTVD(img.shape, img.dtype, promotion="numpy2") as st:
            st.set_coeffs(kw["mu"], kw["omega"], kw["ell"], img)
            st.upload(img)
            r = st.run(n, jacobi_maxiter=sweeps)
            pipelined = shape[-1] in (1, 3) or len(shape) == 2
            assert (r.launches == n) == (tune is None and pipelined), (r.launches, chunks, tune)
            u, d, b = st.download(want_split=True)
            bits(u, ref)
            bits(d, np.stack(dl, axis=-1))
            bits(b, np.stack(bl, axis=-1))
            r = st.run(3, jacobi_maxiter=sweeps)     # the context goes on from its float64 iterate: n + 3 iterations in all
            bits(st.download(), ref2)
    monkeypatch.delenv("DXH_NCHUNKS", raising=False)
    monkeypatch.delenv("DXH_FUSED_TUNE", raising=False)
    bits(da.split_bregman_tvd(img, max_num_iter=n, solver=da.Jacobi(maxiter=sweeps), **kw), ref)


# ---- volumes with several Jacobi sweeps: two or three sweeps per pass in plane tiles (dxh_vol_multi.h) ----
@pytest.mark.parametrize("sweeps", [2, 3, 4, 5, 7, 20])
@pytest.mark.parametrize("shape", [(9, 40, 70), (30, 16, 32), (5, 61, 129)])
@pytest.mark.parametrize("kind", ["float32 numpy2", "float32 native", "float64"])
@pytest.mark.parametrize("omega_kind", ["scalar omega", "array omega"])
def test_volume_sweeps_in_plane_tiles_bit_for_bit_against_the_oracle(sweeps, shape, kind, omega_kind, monkeypatch):
    """The reference's own volume workloads run Jacobi(maxiter=20) inside the TV loop (examples/regularization.py:8-31,
    examples/paper/pet_simulations_comparison_block_b.py:700-713).  Tiles that touch every face of the volume and interior
    ones, plane chunks that start inside the volume, every grouping of a sweep count into passes of three and two (+ a
    single) -- against the oracle (jacobi.py:107-114 with replicate ghosts along all three axes), and against the one-sweep
    kernels (DXH_FUSED_TUNE bit 67108864), whose launch count shows that the passes replaced them."""
    rng = np.random.default_rng(sweeps * 31 + shape[2])
    dtype = np.float64 if kind == "float64" else np.float32
    vol = rng.random(shape).astype(dtype)
    promotion = "native" if kind.endswith("native") else ("numpy2" if dtype == np.float32 else None)
    kw = dict(mu=0.2, omega=0.8, ell=0.4)
    if omega_kind == "array omega":   # the sweeps' divisor becomes a plane (pet_simulations_comparison_block_b.py:640-672: omega per voxel)
        kw["omega"] = (0.5 + rng.random(shape)).astype(np.float64 if promotion == "numpy2" else dtype)
    n = 3
    ref = O.split_bregman_tvd(vol, dim=3, max_num_iter=n, jacobi_maxiter=sweeps,
                              promotion={"native": "strict", "numpy2": "numpy2", None: "strict"}[promotion], **kw)
    launches = {}
    for tune in (None, "67108864"):
        monkeypatch.delenv("DXH_FUSED_TUNE", raising=False)
        if tune:
            monkeypatch.setenv("DXH_FUSED_TUNE", tune)
        pk = {} if promotion is None else dict(promotion=promotion)
        with da.DeviceTVD(vol.shape, vol.dtype, dim=3, **pk) as st:
            st.set_coeffs(kw["mu"], kw["omega"], kw["ell"], vol)
            st.upload(vol)
            launches[tune] = st.run(n, jacobi_maxiter=sweeps).launches
            bits(st.download(), ref)
    # (numpy2: the first sweep of an iteration belongs to the fused shrinkage / right-hand-side pass or sums in float32)
    if sweeps - (1 if promotion == "numpy2" and omega_kind == "scalar omega" else 0) >= 2:
        assert launches[None] < launches["67108864"], launches
    monkeypatch.delenv("DXH_FUSED_TUNE", raising=False)


@pytest.mark.parametrize("sweeps", [2, 3, 4, 5, 6])
@pytest.mark.parametrize("shape", [(41, 700, 3), (300, 517), (23, 60, 3), (2, 2)])
def test_float64_images_with_several_sweeps_run_as_one_pass_per_iteration(sweeps, shape, monkeypatch):
    """float64 images (what scikit-image's img_as_float hands the reference's pipelines) with Jacobi(maxiter = 2 .. 6): the
    sweeps pipeline with every quantity float64 (dxh_np2_sweeps.h, F = double; the tail repeats the head's shrinkage of the
    stored q instead of holding b in LDS) -- against the oracle, with the launch count, and against the three launches per
    iteration it replaces (DXH_FUSED_TUNE bit 33554432)."""
    rng = np.random.default_rng(sweeps * 7 + shape[1])
    img = rng.random(shape)
    kw = dict(mu=0.13, omega=0.9, ell=0.35)
    n = 4
    ref, dl, bl, _ = O.split_bregman_tvd(img, max_num_iter=n, jacobi_maxiter=sweeps, return_state=True, **kw)
    ref2 = O.split_bregman_tvd(img, max_num_iter=n + 3, jacobi_maxiter=sweeps, **kw)
    for chunks, tune in ((None, None), ("2", None), (None, "33554432")):
        if chunks and shape[0] < 200:
            continue
        monkeypatch.delenv("DXH_NCHUNKS", raising=False)
        monkeypatch.delenv("DXH_FUSED_TUNE", raising=False)
        if chunks:
            monkeypatch.setenv("DXH_NCHUNKS", chunks)
        if tune:
            monkeypatch.setenv("DXH_FUSED_TUNE", tune)
        with da.DeviceTVD(img.shape, img.dtype) as st:
            st.set_coeffs(kw["mu"], kw["omega"], kw["ell"], img)
            st.upload(img)
            r = st.run(n, jacobi_maxiter=sweeps)
            assert (r.launches == n) == (tune is None), (r.launches, chunks, tune)
            u, d, b = st.download(want_split=True)
            bits(u, ref)
            bits(d, np.stack(dl, axis=-1))
            bits(b, np.stack(bl, axis=-1))
            st.run(3, jacobi_maxiter=sweeps)
            bits(st.download(), ref2)
    monkeypatch.delenv("DXH_NCHUNKS", raising=False)
    monkeypatch.delenv("DXH_FUSED_TUNE", raising=False)
    bits(da.split_bregman_tvd(img, max_num_iter=n, solver=da.Jacobi(maxiter=sweeps), **kw), ref)
    # with the stopping criterion the same pipeline serves every iteration (the increment is reduced after each pass)
    bits(da.split_bregman_tvd(img, max_num_iter=30, eps=1e-3, solver=da.Jacobi(maxiter=sweeps), **kw),
         O.split_bregman_tvd(img, max_num_iter=30, eps=1e-3, jacobi_maxiter=sweeps, **kw))
