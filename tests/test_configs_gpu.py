"""BASELINE.json configs[2] and configs[3] at their real sizes (`pytest -m gpu`).

configs[2]: colour correction -> isotropic TVD as ONE device-resident chain on a 4000 x 8000 x 3 frame
(reference chaining: image/image.py:188-199, corrections/color/colorcorrection.py:388-394 -> restoration/tvd.py).
configs[3]: DiffeomorphicImageRegistration of two 4000 x 8000 gray frames, N_patches = [10, 20]
(corrections/shape/piecewiseperspective.py:71-133, multi_image_analysis/imageregistration.py:265-306).

Full-size results are checked against the CPU oracles through windows / samples that the oracles finish in
seconds: an n-iteration TV result depends on the input within n + 1 pixels only (tests/test_tvd_gpu.py states
the argument), a warped patch depends on its own record only, a thin-plate-spline value on its point only.
"""
import numpy as np
import pytest

import darsia_amd as da
from darsia_amd.synthetic import synthetic_image
from oracle import tvd_numpy as O
from oracle import warp_numpy as W

pytestmark = pytest.mark.gpu

H, WID = 4000, 8000


def _colour_setup():
    rng = np.random.default_rng(7)                      # A = I + 0.05 R, b = 0.01 r from seed 7
    A = np.eye(3) + 0.05 * rng.normal(size=(3, 3))
    b = 0.01 * rng.normal(size=3)
    frame = (synthetic_image(H, WID, 3) * 255.0 + 0.5).astype(np.uint8)   # a photograph: uint8 RGB
    return frame, A, b


def test_config3_colour_correction_then_isotropic_tvd_chain():
    frame, A, b = _colour_setup()
    n_iter, mu = 40, 0.1
    bal = da.AffineBalance()
    bal.balance_scaling, bal.balance_translation = A, b

    # --- the chain, device-resident: the uint8 frame goes up once, the float32 result comes down once
    d_frame = da.DeviceArray.from_host(frame)
    d_col = bal.apply_balance(d_frame)
    assert isinstance(d_col, da.DeviceArray) and d_col.dtype == np.float32 and d_col.shape == frame.shape
    # (regularization = ell = 2 mu as in split_bregman_tvd's own default; TVD's default of 1.0 makes the reference's
    # channel-coupled Jacobi sweep expansive on colour images -- 6 ell / (omega + 4 ell) > 1, quirk Q2)
    kw = dict(method="heterogeneous bregman", weight=mu, max_num_iter=n_iter, eps=None, isotropic=True,
              regularization=2 * mu)
    d_out = da.tvd(d_col, **kw)
    assert isinstance(d_out, da.DeviceArray)
    out = d_out.get()
    col = d_col.get()

    # --- the same chain hopping through host arrays gives the same bits
    col_h = bal.apply_balance(frame)
    assert np.array_equal(col_h, col)
    out_h = da.tvd(col_h, **kw)
    assert np.array_equal(out_h, out)

    # --- colour stage against the reference expression (colorbalance.py:186-196) on the whole frame; the device sums
    # the three products in index order in double, NumPy's matmul in BLAS order: equal after the float32 cast up to
    # one unit in the last place
    ref_col = ((frame.reshape(-1, 3).astype(np.float64) / 255.0) @ A + b).astype(np.float32).reshape(frame.shape)
    np.testing.assert_allclose(col, ref_col, rtol=2e-7, atol=1e-7)

    # --- TV stage against the oracle on windows (corner: two sides are image boundaries; centre: none)
    m = n_iter + 2
    for (r0, c0) in ((0, 0), ((H - 384) // 2, (WID - 384) // 2), (H - 384, WID - 384)):
        ra, rb = (0 if r0 == 0 else m), (384 if r0 + 384 == H else 384 - m)
        ca, cb = (0 if c0 == 0 else m), (384 if c0 + 384 == WID else 384 - m)
        crop = np.ascontiguousarray(col[r0:r0 + 384, c0:c0 + 384])
        got = out[r0 + ra:r0 + rb, c0 + ca:c0 + cb]
        # bit for bit against the oracle in the image dtype, fed with the device's colour stage
        want = O.split_bregman_tvd(crop, mu=mu, max_num_iter=n_iter, isotropic=True, promotion="strict")[ra:rb, ca:cb]
        assert np.array_equal(got, want), f"window ({r0},{c0}): max abs {np.abs(got - want).max():.3e}"
        # the whole CPU chain (NumPy colour stage -> oracle in the reference's dtype flow): north-star tolerance
        crop_ref = np.ascontiguousarray(ref_col[r0:r0 + 384, c0:c0 + 384])
        want2 = O.split_bregman_tvd(crop_ref, mu=mu, max_num_iter=n_iter, isotropic=True)[ra:rb, ca:cb]
        np.testing.assert_allclose(got, want2, rtol=1e-4, atol=1e-6)
    # full-size invariants
    assert np.isfinite(out).all()
    tv = lambda a: np.abs(np.diff(a, axis=0)).sum(dtype=np.float64) + np.abs(np.diff(a, axis=1)).sum(dtype=np.float64)  # noqa: E731
    assert tv(out) < tv(col)


def _frames():
    from scipy.ndimage import map_coordinates

    dst = synthetic_image(H, WID, 1)
    yy, xx = np.mgrid[0:H, 0:WID].astype(np.float32)
    dx = 4.0 * np.sin(np.pi * yy / H)      # (4 sin(pi y / H), 6 sin(pi x / W)) pixels
    dy = 6.0 * np.sin(np.pi * xx / WID)
    src = map_coordinates(dst, [yy + dy, xx + dx], order=1, mode="nearest").astype(np.float32)
    return dst, src


def test_config4_registration_of_two_full_size_frames():
    from scipy.interpolate import RBFInterpolator

    dst, src = _frames()
    npatch = [10, 20]

    # --- the dense warp at size: every sampled patch of the device canvas equals the oracle's warp of that patch
    def disp(pts_xy):
        x, y = pts_xy[:, 0], pts_xy[:, 1]
        return np.array([4.0 * np.sin(np.pi * y / H), 6.0 * np.sin(np.pi * x / WID)])

    patches = da.Patches(src, npatch, rel_overlap=0.0)
    out = da.PiecewisePerspectiveTransform().find_and_warp(patches, disp, reverse=True)
    assert out.img.shape == src.shape and out.img.dtype == src.dtype
    for (i, j) in ((0, 0), (0, 19), (9, 0), (9, 19), (4, 10), (5, 3), (2, 17)):
        ref = W.find_and_warp(src, [[patches.rois[i][j]]], patches.local_corners_voxels[i:i + 1, j:j + 1],
                              patches.global_corners_voxels[i:i + 1, j:j + 1], disp, True)
        r, c = patches.rois[i][j]
        # inside the patch, away from the seams it shares with its neighbours (those rows / columns are sums, quirk Q8)
        rs, cs = slice(r.start + 12, r.stop - 12), slice(c.start + 12, c.stop - 12)
        assert np.array_equal(out.img[rs, cs], ref[rs, cs]), (i, j, np.abs(out.img[rs, cs] - ref[rs, cs]).max())

    # --- the registration facade on the two frames: the estimated map pulls src back onto dst
    reg = da.ImageRegistration(da.Image(dst), N_patches=npatch)
    aligned = reg(da.Image(src))
    # (measured on low-passed differences: the pixel noise of the synthetic frames is white, so resampling alone
    # changes it as much as a shift does; the misplaced block edges of the pattern are what alignment removes)
    from scipy.ndimage import gaussian_filter

    inner = (slice(400, 1900), slice(400, 3400))
    before = np.abs(gaussian_filter((src - dst)[inner], 3.0)).mean()
    after = np.abs(gaussian_filter((aligned.img - dst)[inner], 3.0)).mean()
    assert after < 0.6 * before, (before, after)

    # --- the dense displacement field (all 32 M pixels on the device) against SciPy on a sub-grid
    ta = reg.registration.translation_analysis
    field = reg.displacement()
    assert field.shape == (H * WID, 2)
    rows = np.arange(0, H, 397)
    cols = np.arange(0, WID, 401)
    rr, cc = np.meshgrid(rows, cols, indexing="ij")
    pts = np.c_[cc.ravel(), rr.ravel()].astype(np.float64)          # (x, y) pixels
    tx = ta.interpolator_translation_x(pts)
    ty = ta.interpolator_translation_y(pts)
    want = ta.base.coordinatesystem.coordinate_vector(np.transpose(np.vstack((ty, tx))))
    got = field.reshape(H, WID, 2)[rr.ravel(), cc.ravel()]
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12)
    assert isinstance(ta.interpolator_translation_x, RBFInterpolator)


def test_resize_tvd_resize_chain_stays_on_the_device():
    """The presets' "scale by downsampling" chain, CombinedModel([Resize, TVD, Resize])
    (presets/fluidflower/benchmarkco2model.py:78-84, signals/models/combinedmodel.py:29-52): given a DeviceArray it
    runs without a host hop and returns a DeviceArray whose bits equal the chain through host arrays."""
    sig = synthetic_image(1500, 2200, 1)
    chain = da.CombinedModel([
        da.Resize(fx=0.5, fy=0.5, interpolation="inter_area"),
        da.TVD(method="heterogeneous bregman", weight=0.05, regularization=0.1, max_num_iter=30, eps=None),
        da.Resize(shape=sig.shape, interpolation="inter_linear"),
    ])
    host = chain(sig)
    dev = chain(da.DeviceArray.from_host(sig))
    assert isinstance(dev, da.DeviceArray) and dev.shape == sig.shape and dev.dtype == sig.dtype
    assert np.array_equal(dev.get(), host)
    # the default TVD() method (Chambolle) chains the same way
    chain2 = da.CombinedModel([da.Resize(fx=0.5, fy=0.5, interpolation="inter_area"), da.TVD(weight=0.1, max_num_iter=20)])
    assert np.array_equal(chain2(da.DeviceArray.from_host(sig)).get(), chain2(sig))


def test_whole_frame_pipeline_stays_in_hbm():
    """A raw uint8 frame through colour correction, curvature correction, registration onto the baseline and concentration
    analysis: one upload, one download, every stage reads and writes device arrays -- and the result equals the same chain run
    operator by operator through host arrays, bit for bit."""
    import warnings

    from scipy.ndimage import gaussian_filter, map_coordinates

    rng = np.random.default_rng(77)
    h, w = 240, 352
    tex = np.stack([gaussian_filter(rng.random((h, w)), 2.5) for _ in range(3)], axis=-1)
    tex = (tex - tex.min()) / (tex.max() - tex.min())
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    field = np.sin(np.pi * yy / h) * np.sin(np.pi * xx / w)
    moved = np.stack([map_coordinates(tex[..., c], [yy - 1.2 * field, xx + 2.0 * field], order=1, mode="nearest") for c in range(3)], axis=-1)
    blob = np.exp(-(((yy - h / 2) / (h / 6)) ** 2 + ((xx - w / 3) / (w / 7)) ** 2))
    raw_base = (255 * np.clip(0.6 * tex + 0.1, 0, 1)).astype(np.uint8)
    raw_frame = (255 * np.clip(0.6 * moved + 0.1 + 0.25 * blob[..., None] * np.array([1.0, 0.3, 0.1]), 0, 1)).astype(np.uint8)

    balance = da.AffineBalance()
    balance.balance_scaling, balance.balance_translation = np.eye(3) + 0.03 * rng.random((3, 3)), 0.005 * rng.random(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        curvature = da.CurvatureCorrection(config={"bulge": dict(horizontal_bulge=2e-6, vertical_bulge=-1e-6)})
    tvd = da.TVD(method="heterogeneous bregman", weight=0.05, omega=1.0, regularization=0.1, max_num_iter=10, eps=None)
    model = da.CombinedModel([da.LinearModel(scaling=2.0, offset=0.0), da.ClipModel(min_value=0.0, max_value=1.0)])

    def prepare(raw):                       # uint8 (host or device) -> corrected float32 frame of the same kind
        return curvature(balance.apply_balance(raw))

    # operator by operator through host arrays
    base_h, frame_h = da.OpticalImage(prepare(raw_base)), da.OpticalImage(prepare(raw_frame))
    reg_h = da.ImageRegistration(base_h, N_patches=[3, 4], rel_overlap=0.1)
    aligned_h = reg_h(frame_h)
    ca = da.ConcentrationAnalysis(base_h, da.MonochromaticReduction(color="red"), None, tvd, model, **{"diff option": "positive"})
    want = ca(da.OpticalImage(aligned_h.img))
    # the same with the pixels in HBM from the first stage on
    base_d, frame_d = prepare(da.DeviceArray.from_host(raw_base)), prepare(da.DeviceArray.from_host(raw_frame))
    assert isinstance(frame_d, da.DeviceArray) and np.array_equal(frame_d.get(), frame_h.img)
    reg_d = da.ImageRegistration(da.OpticalImage(base_d), N_patches=[3, 4], rel_overlap=0.1)
    aligned_d = reg_d(da.OpticalImage(frame_d))
    assert isinstance(aligned_d.img, da.DeviceArray) and np.array_equal(aligned_d.img.get(), aligned_h.img)
    got = ca(da.OpticalImage(aligned_d.img))
    assert isinstance(got.img, da.DeviceArray) and np.array_equal(got.img.get(), want.img)
    # an analysis built on the device-resident baseline is the same analysis
    ca_d = da.ConcentrationAnalysis(da.OpticalImage(base_d), da.MonochromaticReduction(color="red"), None, tvd, model, **{"diff option": "positive"})
    assert isinstance(ca_d.base.img, np.ndarray) and np.array_equal(ca_d(da.OpticalImage(aligned_d.img)).img.get(), want.img)
    assert want.img.max() > 0.2            # the tracer blob is there
