"""The oracle's numpy2 flow against vectors produced by executing the reference on float32 images outside the default
solve: several Jacobi sweeps, volumes (dim = 3), six channels (tests/golden/make_golden_np2_generic.py).  Bit for bit."""
import os

import numpy as np
import pytest

from oracle import tvd_numpy as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "n1_gray_m2": dict(mu=0.1, max_num_iter=12),
    "n2_gray_m3_mu1": dict(mu=1.0, omega=0.7, ell=0.45, max_num_iter=15),
    "n3_rgb_m5": dict(mu=0.1, max_num_iter=10),
    "n4_rgb_m6_ell": dict(mu=0.2, ell=0.3, max_num_iter=8),
    "n5_gray_m4_muarr": dict(mu="arr", ell=0.5, max_num_iter=10),
    "n6_vol_m1": dict(mu=0.1, dim=3, max_num_iter=10),
    "n7_vol_m4": dict(mu=0.3, omega=0.5, ell=0.4, dim=3, max_num_iter=8),
    "n8_vol_m2_muarr": dict(mu="arr", ell=0.6, dim=3, max_num_iter=8),
    "n9_vol_m7": dict(mu=0.1, dim=3, max_num_iter=5),
    "n10_six_m1": dict(mu=0.1, max_num_iter=9),
    "n11_six_m3": dict(mu=0.15, ell=0.5, max_num_iter=7),
}


@pytest.fixture(scope="module")
def generic():
    return np.load(os.path.join(ROOT, "tests", "golden", "tvd_np2_generic_golden.npz"), allow_pickle=False)


def case_kwargs(generic, name):
    kw = dict(CASES[name])
    if isinstance(kw.get("mu"), str):
        kw["mu"] = generic[name + "/mu"]
    return kw, int(generic[name + "/maxiter"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_the_reference_bit_for_bit(generic, name):
    kw, m = case_kwargs(generic, name)
    img = generic[name + "/img"]
    assert img.dtype == np.float32
    out = O.split_bregman_tvd(img, jacobi_maxiter=m, promotion="numpy2", **kw)
    ref = generic[name + "/out"]
    assert out.dtype == ref.dtype == np.float32
    assert np.array_equal(out, ref)


def test_oracle_stop_iteration_with_several_sweeps(generic):
    img = generic["n12_gray_m3_eps/img"]
    out, _, _, done = O.split_bregman_tvd(img, mu=0.1, max_num_iter=200, eps=2e-3, jacobi_maxiter=3, promotion="numpy2",
                                          return_state=True)
    assert done == int(generic["n12_gray_m3_eps/iters"])
    assert np.array_equal(out, generic["n12_gray_m3_eps/out"])


TOL_CASES = {
    "t1_gray_tol": dict(m=20, tol=1e-3, kw=dict(mu=0.1, max_num_iter=6)),
    "t2_rgb_tol": dict(m=15, tol=3e-3, kw=dict(mu=0.2, ell=0.5, max_num_iter=5)),
    "t3_vol_tol": dict(m=12, tol=2e-3, kw=dict(mu=0.1, dim=3, max_num_iter=4)),
    "t4_gray_tol_loose": dict(m=5, tol=0.5, kw=dict(mu=0.1, max_num_iter=3)),
}
JACOBI_CASES = {
    "j1_gray_m1": dict(m=1, tol=None, mass=1.0, diff=0.2, dim=2),
    "j2_gray_m7": dict(m=7, tol=None, mass=0.7, diff=0.35, dim=2),
    "j3_rgb_m4": dict(m=4, tol=None, mass=1.0, diff=0.2, dim=2),
    "j4_vol_m3": dict(m=3, tol=None, mass=1.3, diff=0.4, dim=3),
    "j5_gray_tol": dict(m=30, tol=1e-3, mass=1.0, diff=0.2, dim=2),
}
H1_CASES = {
    "h1_gray_m4": dict(m=4, tol=None, kw=dict(mu=0.4, omega=1.3, dim=2)),
    "h2_rgb_m4": dict(m=4, tol=None, kw=dict(mu=0.4, omega=1.3, dim=2)),
    "h3_rgb_tol": dict(m=30, tol=1e-3, kw=dict(mu=1.0, omega=1.0, dim=2)),
    "h4_vol_m6": dict(m=6, tol=None, kw=dict(mu=0.5, omega=0.8, dim=3)),
    "h5_gray_m1": dict(m=1, tol=None, kw=dict(mu=1.0, omega=1.0, dim=2)),
}


@pytest.mark.parametrize("name", sorted(TOL_CASES))
def test_oracle_with_a_tolerance_in_the_jacobi_solver(generic, name):
    c = TOL_CASES[name]
    out = O.split_bregman_tvd(generic[name + "/img"], jacobi_maxiter=c["m"], jacobi_tol=c["tol"], promotion="numpy2", **c["kw"])
    assert out.dtype == np.float32 and np.array_equal(out, generic[name + "/out"])


@pytest.mark.parametrize("name", sorted(JACOBI_CASES))
def test_oracle_jacobi_on_float32_operands_returns_float64_like_the_reference(generic, name):
    c = JACOBI_CASES[name]
    out = O.jacobi(generic[name + "/x0"], generic[name + "/rhs"], c["mass"], c["diff"], dim=c["dim"], maxiter=c["m"], tol=c["tol"],
                   promotion="numpy2")
    ref = generic[name + "/out"]
    assert ref.dtype == np.float64 and out.dtype == np.float64 and np.array_equal(out, ref)


@pytest.mark.parametrize("name", sorted(H1_CASES))
def test_oracle_h1_regularization_of_float32_images(generic, name):
    c = H1_CASES[name]
    out = O.h1_regularization(generic[name + "/img"], jacobi_maxiter=c["m"], jacobi_tol=c["tol"], promotion="numpy2", **c["kw"])
    ref = generic[name + "/out"]
    assert ref.dtype == np.float32 and out.dtype == np.float32 and np.array_equal(out, ref)


ARRAY_CASES = {   # name -> scalar keyword arguments (arrays come from the file)
    "a1_gray_omega": dict(mu=0.1, max_num_iter=10),
    "a2_gray_omega_ell": dict(mu=0.1, ell=0.4, max_num_iter=10),
    "a3_gray_ellarr": dict(mu=0.1, max_num_iter=10),
    "a4_rgb_all": dict(max_num_iter=8),
    "a5_vol_munone": dict(dim=3, max_num_iter=6),
    "a6_gray_omega_m3": dict(ell=0.5, max_num_iter=7),
    "a7_vol_ellarr_m2": dict(mu=0.2, omega=0.7, dim=3, max_num_iter=5),
    "f1_gray_munone_f32": dict(max_num_iter=10),
    "f2_rgb_omega_f32": dict(mu=0.1, max_num_iter=8),
    "f3_gray_all_f32_m2": dict(max_num_iter=6),
}


def array_kwargs(generic, name):
    kw = dict(ARRAY_CASES[name])
    for k in ("mu", "omega", "ell"):
        if f"{name}/{k}" in generic.files:
            kw[k] = generic[f"{name}/{k}"]
    return kw, int(generic[name + "/maxiter"])


@pytest.mark.parametrize("name", sorted(ARRAY_CASES))
def test_oracle_with_array_valued_omega_and_ell(generic, name):
    kw, m = array_kwargs(generic, name)
    out = O.split_bregman_tvd(generic[name + "/img"], jacobi_maxiter=m, promotion="numpy2", **kw)
    assert out.dtype == np.float32 and np.array_equal(out, generic[name + "/out"])
    if name.startswith("f"):   # float32 arrays only: the reference's flow is float32 throughout -- the same-dtype flow
        assert np.array_equal(O.split_bregman_tvd(generic[name + "/img"], jacobi_maxiter=m, promotion="strict", **kw), generic[name + "/out"])


WARM_CASES = {
    "w1_gray": dict(mu=0.1),
    "w2_rgb_muarr_m2": dict(ell=0.5),
    "w3_vol_u0_f64": dict(mu=0.2, omega=0.7, ell=0.4, dim=3),
    "w4_gray_one": dict(mu=1.0, ell=0.45),
    # next to array-valued omega / ell (round 4)
    "w5_gray_omega_ell_arr_m2": dict(mu=0.1),
    "w6_vol_ellarr": dict(mu=0.2, omega=0.7, dim=3),
    "w7_rgb_munone_u0_f64": dict(),
}


def warm_inputs(generic, name):
    kw = dict(WARM_CASES[name])
    for c in ("mu", "omega", "ell"):
        if f"{name}/{c}" in generic.files:
            kw[c] = generic[f"{name}/{c}"]
    return kw, int(generic[name + "/maxiter"]), int(generic[name + "/iters"])


@pytest.mark.parametrize("name", sorted(WARM_CASES))
def test_oracle_warm_start_of_a_float32_image(generic, name):
    kw, m, n = warm_inputs(generic, name)
    dim = kw.get("dim", 2)
    d0, b0 = generic[name + "/d0"], generic[name + "/b0"]
    x0 = (generic[name + "/u0"], [d0[..., j].copy() for j in range(dim)], [b0[..., j].copy() for j in range(dim)])
    out, d, b, _ = O.split_bregman_tvd(generic[name + "/img"], x0=x0, jacobi_maxiter=m, max_num_iter=n, promotion="numpy2",
                                       return_state=True, **kw)
    assert np.array_equal(out, generic[name + "/out"])
    assert np.array_equal(np.stack(d, axis=-1), generic[name + "/d"]) and np.array_equal(np.stack(b, axis=-1), generic[name + "/b"])


ADAPT_CASES = {
    "d1_gray": dict(mu=0.1, max_num_iter=6),
    "d2_rgb_muarr": dict(omega=0.8, ell=0.6, max_num_iter=5),
    "d3_vol_m2": dict(mu=0.2, omega=0.7, ell=0.4, dim=3, max_num_iter=4),
    "d4_gray_late": dict(mu=1.0, ell=0.45, max_num_iter=6),
}


def adapt_inputs(generic, name):
    kw = dict(ADAPT_CASES[name])
    if f"{name}/mu" in generic.files:
        kw["mu"] = generic[f"{name}/mu"]
    when = set(int(v) for v in generic[name + "/when"])
    return kw, int(generic[name + "/maxiter"]), (lambda it: it in when)


@pytest.mark.parametrize("name", sorted(ADAPT_CASES))
def test_oracle_adaptive_regularisation_of_float32_images(generic, name):
    kw, m, sched = adapt_inputs(generic, name)
    out = O.split_bregman_tvd(generic[name + "/img"], jacobi_maxiter=m, adaptive=sched, promotion="numpy2", **kw)
    assert out.dtype == np.float32 and np.array_equal(out, generic[name + "/out"])
