"""CPU-side checks of the C ABI: the library loads, exports every symbol the
header declares, reports errors without a GPU, and the host layer mirrors the
reference's argument handling.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import darsia_amd as da
from darsia_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "darsia_hip.h")).read()
    declared = set(re.findall(r"\b(dxh_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("dxh_tvd_desc")
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dxh_version() >= 100


def test_desc_struct_matches_header_layout():
    # int32 x6, int64 x3, int32 x2, pointer -> 64 bytes on LP64
    assert C.sizeof(_lib.TvdDesc) == 64
    assert C.sizeof(_lib.TvdResult) == 32


def test_bad_arguments_are_reported_not_aborted():
    lib = _lib.load()
    h = C.c_void_p()
    d = _lib.TvdDesc()
    d.struct_size = 3  # wrong on purpose
    assert lib.dxh_tvd_create(C.byref(h), C.byref(d)) == -1
    assert b"size mismatch" in lib.dxh_last_error()
    with pytest.raises(ValueError):
        _lib.check(lib.dxh_tvd_run(None, 1, 1, -1.0, -1.0, 0.0, 0, None))


@pytest.mark.skipif(_lib.device_count() > 0, reason="GPU present")
def test_fails_loudly_without_gpu():
    with pytest.raises(da.HipError):
        da.split_bregman_tvd(np.zeros((8, 8), np.float32), mu=0.1, max_num_iter=1)
    with pytest.raises(da.HipError):
        da.Jacobi(mass_coeff=1.0, diffusion_coeff=1.0)(np.zeros((4, 4)), np.zeros((4, 4)))
    with pytest.raises(da.HipError):
        da.backward_diff(np.zeros((4, 4)), 0)
    with pytest.raises(da.HipError):
        da.DeviceArray((4, 4), np.float32)
    with pytest.raises(da.HipError):
        da.DeviceArray.from_host(np.zeros((4, 4), np.float32))
    from darsia_amd.parallel import Comm

    with pytest.raises(da.HipError):
        Comm.rccl(0, 1, device=0)


def test_tvd_kwarg_parsing_matches_reference(golden):
    """restoration/tvd.py:23-43 -- values captured from the reference in tests/golden."""
    t = da.TVD(key="restoration ", **{
        "restoration method": "Heterogeneous Bregman", "restoration weight": 0.3,
        "restoration max_num_iter": 17, "restoration eps": 1e-5, "omega": 2.0, "dim": 2})
    got = repr((t.method, t.weight, t.max_num_iter, t.eps, t.omega, t.regularization, sorted(t.kwargs)))
    assert got == str(golden["meta/g13_attrs"])
    t2 = da.TVD()
    assert repr((t2.method, t2.weight, t2.max_num_iter, t2.eps, sorted(t2.kwargs))) == str(golden["meta/g13_defaults"])
    # Q4 divergence: 'regularization' is consumed instead of being forwarded twice
    t4 = da.TVD(method="heterogeneous bregman", regularization=0.5)
    assert t4.regularization == 0.5 and "regularization" not in t4.kwargs


def test_error_types_match_reference(golden):
    with pytest.raises(ValueError, match="Method nonsense not supported."):
        da.TVD(method="nonsense")(np.zeros((4, 4), np.float32))
    assert str(golden["meta/g13_bad_method"]) == "ValueError: Method nonsense not supported."
    with pytest.raises(TypeError, match="not supported"):
        da.TVD()([1, 2, 3])
    with pytest.raises(TypeError):
        da.split_bregman_tvd(np.zeros((4, 4), np.uint8))
    with pytest.raises(TypeError):
        da.TVD(method="isotropic bregman")(np.zeros((4, 4), np.complex64))  # img_as_float rejects it


def test_operator_call_signature_is_plain_method():
    """CombinedModel inspects ``model.__call__.__code__.co_argcount == 2``
    (signals/models/combinedmodel.py:29-52)."""
    assert da.TVD.__call__.__code__.co_argcount == 2


def test_solver_update_params_semantics():
    s = da.Jacobi(maxiter=3, tol=1e-3, dim=2, mass_coeff=1.0, diffusion_coeff=0.5)
    s.update_params(mass_coeff=2.0)
    assert (s.mass_coeff, s.diffusion_coeff, s.dim, s.maxiter, s.tol) == (2.0, 0.5, 2, 3, 1e-3)
    s.update_params(dim=3, diffusion_coeff=0.1)
    assert (s.dim, s.diffusion_coeff) == (3, 0.1)


def test_image_container_copy_is_deep():
    im = da.Image(np.arange(6.0).reshape(2, 3), name="x", tag="frame 7")
    cp = im.copy()
    cp.img[0, 0] = 99
    assert im.img[0, 0] == 0 and cp.metadata()["name"] == "x" and cp.metadata()["tag"] == "frame 7"
    again = da.Image(cp.img, **cp.metadata())
    assert again.dimensions == im.dimensions and np.array_equal(again.origin, im.origin)


_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_sources_do_not_import_torch():
    """North star: no PyTorch in the product or the benchmark process (torch.distributed.run is only the launcher)."""
    import re

    pat = re.compile(r"^\s*(import torch|from torch)", re.M)
    offenders = []
    for base, _, files in os.walk(os.path.join(_ROOT, "darsia_amd")):
        for f in files:
            if f.endswith(".py") and pat.search(open(os.path.join(base, f)).read()):
                offenders.append(os.path.join(base, f))
    if pat.search(open(os.path.join(_ROOT, "bench.py")).read()):
        offenders.append("bench.py")
    assert not offenders, offenders
