"""ctypes binding of ``libdarsia_hip.so`` (C ABI: ``include/darsia_hip.h``).

There is NO CPU fallback: if the library is missing, or no MI355X is visible,
every operator raises.  The library is built in-tree by
``darsia_amd/csrc/build.sh`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DARSIA_HIP_LIB") or os.path.join(_HERE, "libdarsia_hip.so")  # env: developer A/B of builds

DXH_F32, DXH_F64 = 0, 1
NSUM_ALL_AXES, NSUM_SPATIAL = 0, 1
FLAG_ISOTROPIC, FLAG_NO_FUSED, FLAG_NO_DPP, FLAG_ASYNC, FLAG_EXPLICIT = 1, 2, 4, 8, 16
FLAG_NO_OVERLAP, FLAG_CONTINUE = 32, 64
COMM_ID_BYTES = 128
SOLVER_JACOBI, SOLVER_MG, SOLVER_CG = 0, 1, 2

# every symbol include/darsia_hip.h declares (tests check that the library exports them all)
EXPORTS = (
    "dxh_last_error", "dxh_version", "dxh_device_count",
    "dxh_tvd_create", "dxh_tvd_destroy", "dxh_tvd_set_coeffs", "dxh_tvd_upload", "dxh_tvd_run",
    "dxh_tvd_download", "dxh_tvd_halo_bytes", "dxh_tvd_pack_rows", "dxh_tvd_unpack_rows",
    "dxh_tvd_partial_norms", "dxh_tvd_sync", "dxh_jacobi_solve", "dxh_diff", "dxh_chambolle",
    "dxh_warp_patches", "dxh_tps_eval", "dxh_affine_color",
    "dxh_solver_create", "dxh_solver_destroy", "dxh_solver_set_params", "dxh_solver_solve", "dxh_solver_get_params",
    "dxh_tvd_set_solver", "dxh_tv_bregman_gs", "dxh_map_coordinates", "dxh_resize", "dxh_div3_probe",
    "dxh_tvd_energy", "dxh_tvd_adapt_ell", "dxh_tvd_set_base", "dxh_tvd_set_cleaning_filter", "dxh_tvd_upload_signal", "dxh_tvd_download_model", "dxh_tvd_set_base_typed", "dxh_tvd_upload_signal_typed", "dxh_warp_patches_typed", "dxh_phase_correlation",
    "dxh_comm_unique_id", "dxh_comm_create_rccl", "dxh_comm_create_host", "dxh_comm_destroy", "dxh_comm_allreduce",
    "dxh_comm_barrier", "dxh_comm_info", "dxh_tvd_attach_comm", "dxh_tvd_exchange", "dxh_tvd_run_slab",
    "dxh_mem_alloc", "dxh_mem_free", "dxh_mem_copy", "dxh_sqrt_probe", "dxh_rcp_probe", "dxh_tvd_solve_banded", "dxh_tvd_solve_banded_typed", "dxh_warp_div_probe", "dxh_jacobi_solve_np2", "dxh_tvd_upload_warm_np2", "dxh_tvd_download_split_np2", "dxh_mem_trim", "dxh_tps_eval_pair", "dxh_tvd_set_coeffs_typed", "dxh_tvd_slab_stats",
    "dxh_phase_correlation_q", "dxh_scale", "dxh_isofac_probe", "dxh_tvd_set_promotion", "dxh_tvd_upload_typed", "dxh_tvd_download_typed",
)

# host-staged transport callbacks (include/darsia_hip.h: dxh_comm_exchange_fn, dxh_comm_allreduce_fn)
COMM_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                               C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
COMM_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)


class HipError(RuntimeError):
    """An error reported by libdarsia_hip (HIP runtime failure, allocation failure, bad state)."""


class TvdDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("dtype", C.c_int32), ("ndim", C.c_int32),
        ("dim", C.c_int32), ("nsum", C.c_int32),
        ("n0", C.c_int64), ("n1", C.c_int64), ("n2", C.c_int64),
        ("top_is_boundary", C.c_int32), ("bottom_is_boundary", C.c_int32),
        ("stream", C.c_void_p),
    ]


class SolverDesc(C.Structure):
    """dxh_solver_desc (include/darsia_hip.h)."""
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("dtype", C.c_int32), ("dim", C.c_int32),
        ("n0", C.c_int64), ("n1", C.c_int64), ("n2", C.c_int64),
        ("kind", C.c_int32), ("depth", C.c_int32), ("smoother_iterations", C.c_int32), ("maxiter", C.c_int32),
        ("heterogeneous", C.c_int32), ("reserved", C.c_int32),
        ("tol", C.c_double), ("h", C.c_double), ("stream", C.c_void_p),
    ]


class ResizePlan(C.Structure):
    """dxh_resize_plan (include/darsia_hip.h)."""
    _fields_ = [
        ("struct_size", C.c_int32), ("mode", C.c_int32), ("isx", C.c_int32), ("isy", C.c_int32),
        ("xi", C.c_void_p), ("xw", C.c_void_p), ("yi", C.c_void_p), ("yw", C.c_void_p),
        ("xptr", C.c_void_p), ("xsrc", C.c_void_p), ("xalpha", C.c_void_p),
        ("yptr", C.c_void_p), ("ysrc", C.c_void_p), ("yalpha", C.c_void_p),
    ]


class WarpPatch(C.Structure):
    """dxh_warp_patch (include/darsia_hip.h)."""
    _fields_ = [
        ("minv", C.c_double * 9), ("sr0", C.c_int32), ("sc0", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
        ("ox", C.c_int32), ("oy", C.c_int32), ("w_eff", C.c_int32), ("h_eff", C.c_int32),
        ("nx", C.c_int32), ("ny", C.c_int32), ("bw0", C.c_int32), ("pad", C.c_int32),
    ]


class ModelOp(C.Structure):
    """dxh_model_op (include/darsia_hip.h)."""
    _fields_ = [("kind", C.c_int32), ("flags", C.c_int32), ("a", C.c_double), ("b", C.c_double)]


class TvdResult(C.Structure):
    _fields_ = [
        ("iters_done", C.c_int32), ("converged", C.c_int32), ("rel_increment", C.c_double),
        ("loop_ms", C.c_double), ("launches", C.c_int64),
    ]


PROMOTION_NUMPY2, PROMOTION_MU_IS_THRESHOLD = 1, 2   # include/darsia_hip.h: DXH_PROMOTION_*


class SlabStats(C.Structure):
    """include/darsia_hip.h: dxh_slab_stats."""
    _fields_ = [
        ("struct_size", C.c_int32), ("wait_mode", C.c_int32), ("launches", C.c_int64),
        ("signalling_launches", C.c_int64), ("exchanges", C.c_int64), ("exchanges_timed", C.c_int64),
        ("wait_timeouts", C.c_int64), ("wait_signal_ms", C.c_double), ("exchange_ms", C.c_double),
    ]


_lib = None
_lock = threading.Lock()


def _declare(lib):
    vp, i, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    lib.dxh_last_error.restype = C.c_char_p
    lib.dxh_last_error.argtypes = []
    lib.dxh_version.restype = i
    lib.dxh_device_count.argtypes = [C.POINTER(i)]
    lib.dxh_tvd_create.argtypes = [C.POINTER(vp), C.POINTER(TvdDesc)]
    lib.dxh_tvd_destroy.argtypes = [vp]
    lib.dxh_tvd_set_coeffs.argtypes = [vp, dbl, vp, dbl, vp, i, dbl, vp]
    lib.dxh_tvd_set_coeffs_typed.argtypes = [vp, dbl, vp, i, dbl, vp, i, i, dbl, vp, i]
    lib.dxh_tvd_upload.argtypes = [vp, vp, vp, vp, vp]
    lib.dxh_tvd_run.argtypes = [vp, i, i, dbl, dbl, dbl, C.c_uint32, C.POINTER(TvdResult)]
    lib.dxh_tvd_download.argtypes = [vp, vp, vp, vp]
    lib.dxh_tvd_halo_bytes.argtypes = [vp, i64, C.POINTER(C.c_size_t)]
    lib.dxh_tvd_pack_rows.argtypes = [vp, i64, i64, vp]
    lib.dxh_tvd_unpack_rows.argtypes = [vp, i64, i64, vp]
    lib.dxh_tvd_partial_norms.argtypes = [vp, i64, i64, C.POINTER(dbl), C.POINTER(dbl)]
    lib.dxh_tvd_sync.argtypes = [vp]
    lib.dxh_div3_probe.argtypes = [i, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_uint)]
    lib.dxh_tvd_energy.argtypes = [vp, i, C.POINTER(dbl)]
    lib.dxh_solver_create.argtypes = [C.POINTER(vp), C.POINTER(SolverDesc)]
    lib.dxh_solver_destroy.argtypes = [vp]
    lib.dxh_solver_set_params.argtypes = [vp, dbl, vp, dbl, vp]
    lib.dxh_solver_solve.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_int32)]
    lib.dxh_solver_get_params.argtypes = [vp, vp, vp]
    lib.dxh_tvd_set_solver.argtypes = [vp, i, i, i, i, dbl]
    lib.dxh_resize.argtypes = [i, i, i64, i64, i, vp, i64, i64, C.POINTER(ResizePlan), vp]
    lib.dxh_map_coordinates.argtypes = [i, i, i64, i64, i, vp, i, vp, vp, i64, i, dbl, vp]
    lib.dxh_tv_bregman_gs.argtypes = [i, i, i64, i64, i64, vp, vp, dbl, i, dbl, i, C.POINTER(i), C.POINTER(dbl)]
    lib.dxh_tvd_adapt_ell.argtypes = [vp]
    lib.dxh_jacobi_solve.argtypes = [i, i, i, i, i, i64, i64, i64, vp, vp, vp, dbl, vp, dbl, vp, dbl, i, dbl,
                                     C.POINTER(i)]
    lib.dxh_tvd_upload_warm_np2.argtypes = [vp, vp, vp, i, vp, vp]
    lib.dxh_tvd_download_split_np2.argtypes = [vp, vp, vp]
    lib.dxh_jacobi_solve_np2.argtypes = [i, i, i, i, i64, i64, i64, vp, vp, vp, i, dbl, dbl, dbl, i, dbl, C.POINTER(i)]
    lib.dxh_diff.argtypes = [i, i, i64, i64, i64, i, i, dbl, vp, vp]
    lib.dxh_chambolle.argtypes = [i, i, i, i64, i64, i64, vp, vp, dbl, dbl, i, C.POINTER(i)]
    lib.dxh_warp_patches.argtypes = [i, i, i, i, vp, vp, i, vp]
    lib.dxh_warp_patches_typed.argtypes = [i, i, i, i, i, vp, vp, i, vp]
    lib.dxh_phase_correlation.argtypes = [i, i, i, i, i, vp, vp, i, vp, dbl, vp, vp]
    lib.dxh_phase_correlation_q.argtypes = [i, i, i, i, i, vp, vp, i, vp, dbl, dbl, vp, vp, vp]
    lib.dxh_tps_eval.argtypes = [i, i, i, vp, vp, vp, vp, vp, i64, vp, i, dbl, dbl, dbl, dbl, vp]
    lib.dxh_tps_eval_pair.argtypes = [i, i, vp, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, i64, i64, dbl, dbl, dbl, dbl, dbl, dbl, vp]
    lib.dxh_affine_color.argtypes = [i, i, i64, vp, vp, vp, i, vp]
    lib.dxh_tvd_set_base.argtypes = [vp, vp, i]
    lib.dxh_tvd_set_cleaning_filter.argtypes = [vp, vp]
    lib.dxh_tvd_upload_signal.argtypes = [vp, vp, i, i, i, vp, i]
    lib.dxh_tvd_set_base_typed.argtypes = [vp, vp, i, i]
    lib.dxh_tvd_upload_signal_typed.argtypes = [vp, vp, i, i, i, i, vp, i]
    lib.dxh_tvd_download_model.argtypes = [vp, vp, vp, i]
    lib.dxh_tvd_solve_banded.argtypes = [C.POINTER(TvdDesc), vp, vp, dbl, vp, dbl, vp, i, dbl, vp, i, i, C.c_uint32, i,
                                         C.POINTER(i)]
    lib.dxh_tvd_solve_banded_typed.argtypes = [C.POINTER(TvdDesc), vp, vp, i, i, dbl, dbl, i, dbl, i, C.c_uint32, i, C.POINTER(i)]
    lib.dxh_warp_div_probe.argtypes = [i, C.c_ulonglong, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.dxh_sqrt_probe.argtypes = [i, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.dxh_rcp_probe.argtypes = [i, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.dxh_scale.argtypes = [i, i, C.c_size_t, vp, dbl, vp]
    lib.dxh_isofac_probe.argtypes = [i, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.dxh_mem_alloc.argtypes = [i, C.c_size_t, C.POINTER(vp)]
    lib.dxh_mem_free.argtypes = [i, vp]
    lib.dxh_mem_trim.argtypes = [C.POINTER(C.c_size_t)]
    lib.dxh_mem_copy.argtypes = [i, vp, vp, C.c_size_t]
    lib.dxh_comm_unique_id.argtypes = [vp]
    lib.dxh_comm_create_rccl.argtypes = [C.POINTER(vp), i, vp, i, i]
    lib.dxh_comm_create_host.argtypes = [C.POINTER(vp), i, i, i, COMM_EXCHANGE_FN, COMM_ALLREDUCE_FN, vp]
    lib.dxh_comm_destroy.argtypes = [vp]
    lib.dxh_comm_allreduce.argtypes = [vp, C.POINTER(dbl), i, i]
    lib.dxh_comm_barrier.argtypes = [vp]
    lib.dxh_comm_info.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i64), C.POINTER(i64)]
    lib.dxh_tvd_attach_comm.argtypes = [vp, vp, i64, i64]
    lib.dxh_tvd_exchange.argtypes = [vp]
    lib.dxh_tvd_run_slab.argtypes = [vp, i, i, dbl, C.c_uint32, C.POINTER(TvdResult)]
    lib.dxh_tvd_slab_stats.argtypes = [vp, C.POINTER(SlabStats)]
    lib.dxh_tvd_set_promotion.argtypes = [vp, i]
    lib.dxh_tvd_upload_typed.argtypes = [vp, vp, i]
    lib.dxh_tvd_download_typed.argtypes = [vp, vp, i]
    for name in EXPORTS:
        if name != "dxh_last_error":
            getattr(lib, name).restype = i


def load():
    """Load the shared library (once).  Raises ImportError when it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} not found: build it with darsia_amd/csrc/build.sh "
                    "(darsia_amd has no CPU fallback)")
            lib = C.CDLL(LIB_PATH)
            _declare(lib)
            _lib = lib
    return _lib


def check(rc: int):
    if rc == 0:
        return
    msg = load().dxh_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    raise HipError(f"libdarsia_hip error {rc}: {msg}")


def device_count() -> int:
    n = C.c_int(0)
    check(load().dxh_device_count(C.byref(n)))
    return n.value


def default_device() -> int:
    return int(os.environ.get("DARSIA_HIP_DEVICE", "0"))


def require_gpu():
    if device_count() < 1:
        raise HipError("no HIP device visible: darsia_amd runs on MI355X only (no CPU fallback)")


def dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.float32:
        return DXH_F32
    if dt == np.float64:
        return DXH_F64
    raise TypeError(f"dtype {dt} not supported: float32/float64 only (SURVEY.md quirk Q5)")


def ptr(a):
    """Pointer to a C-contiguous ndarray or a DeviceArray (or None)."""
    if a is None:
        return None
    if not isinstance(a, np.ndarray):
        return C.c_void_p(a.ptr)   # darsia_amd.device.DeviceArray: dense by construction
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def shape3(shape, dim):
    """(ndim, n0, n1, n2) of an image shape; 2 or 3 axes supported."""
    nd = len(shape)
    if nd not in (2, 3):
        raise ValueError(f"arrays with {nd} axes are not supported (2 or 3 axes)")
    if dim not in (2, 3) or dim > nd:
        raise ValueError(f"dim={dim} not supported for an array with {nd} axes")
    n2 = shape[2] if nd == 3 else 1
    return nd, int(shape[0]), int(shape[1]), int(n2)

