"""ctypes binding of libcaliscope_ba.so (C ABI: include/caliscope_ba.h).

The shared library is built in-tree by ``caliscope_amd/build.py`` (``__graft_entry__.build()``) with
``hipcc --offload-arch=gfx950``.  Loading fails loudly (:class:`BackendError`) when it is missing — there is
no CPU fallback for the arithmetic.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from caliscope_amd.exceptions import BackendError

LIB_NAME = "libcaliscope_ba.so"
LIB_PATH = Path(__file__).resolve().parent / LIB_NAME

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class ProblemDesc(C.Structure):
    _fields_ = [
        ("n_cams", C.c_int32),
        ("n_points", C.c_int32),
        ("n_obs", C.c_int64),
        ("cam_n_params", c_int32_p),
        ("cam_model", c_int32_p),
        ("cam_const", c_double_p),
        ("obs_cam", c_int32_p),
        ("obs_pt", c_int32_p),
        ("obs_uv", c_double_p),
        ("loss", C.c_int32),
        ("f_scale", C.c_double),
    ]


class Options(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("max_blocks", C.c_int32), ("deterministic", C.c_int32), ("evaluation_only", C.c_int32)]


class Linearization(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("g_norm_inf", "gh_sq", "jg_sq", "x_scaled_norm", "x_norm", "cost")]


class NewtonInfo(C.Structure):
    _fields_ = [("ok", C.c_int32), ("reserved", C.c_int32), ("p_sq", C.c_double), ("gh_dot_p", C.c_double), ("w_sq", C.c_double)]


class TrialInfo(C.Structure):
    _fields_ = [("cost", C.c_double), ("step_norm", C.c_double), ("finite", C.c_int32), ("reserved", C.c_int32)]


class StepInfo(C.Structure):
    _fields_ = [
        ("lin", Linearization), ("newton", NewtonInfo), ("trial", TrialInfo), ("lam", C.c_double), ("radius", C.c_double),
        ("p_s", C.c_double * 2), ("predicted", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
        ("need_host", C.c_int32), ("reserved", C.c_int32),
    ]


class Info(C.Structure):
    _fields_ = [
        ("n_cams", C.c_int32), ("n_points", C.c_int32), ("n_cam_params", C.c_int32), ("n_params", C.c_int32),
        ("n_obs", C.c_int64), ("n_chunks", C.c_int32), ("grid_blocks", C.c_int32), ("plan_state", C.c_int32),
        ("max_obs_per_point", C.c_int32), ("device_bytes", C.c_int64),
        ("schur_groups", C.c_int32), ("schur_tiles", C.c_int32), ("schur_grid", C.c_int32), ("n_heavy_points", C.c_int32),
        ("schur_stream_len", C.c_int64), ("schur_pairs", C.c_int64), ("plan_error", C.c_int32), ("build_camg", C.c_int32),
    ]


class SolveOptions(C.Structure):
    _fields_ = [
        ("ftol", C.c_double), ("xtol", C.c_double), ("gtol", C.c_double), ("max_nfev", C.c_int64),
        ("lb", c_double_p), ("ub", c_double_p), ("verbose", C.c_int32), ("max_damping_retries", C.c_int32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("reserved", C.c_int32), ("nfev", C.c_int64), ("njev", C.c_int64), ("n_iterations", C.c_int64),
        ("cost", C.c_double), ("optimality", C.c_double), ("t_total_s", C.c_double),
        ("t_rejected_s", C.c_double), ("n_rejected_timed", C.c_int64),
    ]


class TriangulateDesc(C.Structure):
    _fields_ = [
        ("n_cams", C.c_int32),
        ("cam_model", c_int32_p),
        ("cam_intr", c_double_p),
        ("cam_P", c_double_p),
        ("n_points", C.c_int64),
        ("pt_start", c_int64_p),
        ("obs_cam", c_int32_p),
        ("obs_xy", c_double_p),
        ("float32_io", C.c_int32),
    ]


# name -> (restype, argtypes): every symbol include/caliscope_ba.h declares
SIGNATURES = {
    "cba_create": (C.c_int, [C.POINTER(ProblemDesc), C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "cba_destroy": (None, [C.c_void_p]),
    "cba_comm_unique_id": (C.c_int, [C.c_char_p]),
    "cba_comm_init": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32]),
    "cba_comm_abort": (C.c_int, [C.c_void_p]),
    "cba_group_create": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p)]),
    "cba_group_join": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "cba_group_abort": (None, [C.c_void_p]),
    "cba_group_destroy": (None, [C.c_void_p]),
    "cba_set_constraints": (C.c_int, [C.c_void_p, C.c_int32, c_int32_p, c_int32_p, c_double_p, c_double_p]),
    "cba_begin": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "cba_restart": (C.c_int, [C.c_void_p, c_double_p]),
    "cba_begin_deferred": (C.c_int, [C.c_void_p, c_double_p]),
    "cba_linearize": (C.c_int, [C.c_void_p, C.POINTER(Linearization)]),
    "cba_newton_step": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(NewtonInfo)]),
    "cba_subspace_gram": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, c_double_p]),
    "cba_trial": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.POINTER(TrialInfo)]),
    "cba_accept": (C.c_int, [C.c_void_p]),
    "cba_linearize_build": (C.c_int, [C.c_void_p]),
    "cba_step": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(StepInfo)]),
    "cba_refresh_step_scalars": (C.c_int, [C.c_void_p, C.POINTER(NewtonInfo)]),
    "cba_step_supported": (C.c_int, [C.c_void_p]),
    "cba_set_camera_scaling": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.POINTER(Linearization)]),
    "cba_set_bounds": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "cba_step_camera_state": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "cba_subspace_gram_ex": (C.c_int, [C.c_void_p, C.c_double, C.c_double, c_double_p, C.c_double, C.c_double, c_double_p, c_double_p]),
    "cba_trial_ex": (C.c_int, [C.c_void_p, C.c_double, C.c_double, c_double_p, C.POINTER(TrialInfo)]),
    "cba_solve": (C.c_int, [C.c_void_p, c_double_p, C.POINTER(SolveOptions), c_double_p, C.POINTER(Result)]),
    "cba_get_vector": (C.c_int, [C.c_void_p, C.c_int32, c_double_p]),
    "cba_get_camera_params": (C.c_int, [C.c_void_p, C.c_int32, c_double_p]),
    "cba_get_camera_state": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "cba_residuals": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "cba_normal_blocks": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "cba_reduced_system": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "cba_set_loss": (C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    "cba_plan_wait": (C.c_int, [C.c_void_p]),
    "cba_get_info": (C.c_int, [C.c_void_p, C.POINTER(Info)]),
    "cba_timer_count": (C.c_int, []),
    "cba_timer_name": (C.c_char_p, [C.c_int32]),
    "cba_get_timers": (C.c_int, [C.c_void_p, c_double_p, c_int64_p]),
    "cba_reset_timers": (C.c_int, [C.c_void_p]),
    "cba_enable_timers": (C.c_int, [C.c_void_p, C.c_int32]),
    "cba_host_plan": (C.c_int64, [C.c_int32, C.c_int64, c_int32_p, c_int32_p, C.c_int32, C.c_int32, c_int64_p, c_int64_p, c_int64_p]),
    "cba_triangulate": (C.c_int, [C.POINTER(TriangulateDesc), C.c_int32, c_double_p, c_double_p]),
    "cba_trim": (C.c_int64, []),
    "cba_last_error": (C.c_char_p, []),
    "cba_set_error": (C.c_int, [C.c_int32, C.c_char_p]),
    "cba_version": (C.c_int, []),
    "cba_device_count": (C.c_int, []),
}

_lib = None


def load() -> C.CDLL:
    """Load (once) and type the shared library; raise BackendError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("CALISCOPE_BA_LIB", LIB_PATH))
    if not path.exists():
        raise BackendError(
            f"{path} not found: build the MI355X engine first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or python -m caliscope_amd.build).  There is no CPU fallback."
        )
    try:
        lib = C.CDLL(str(path))
    except OSError as exc:  # missing ROCm runtime etc.
        raise BackendError(f"could not load {path}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise BackendError(f"{path} does not export {name} (stale build?)") from exc
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error(lib) -> str:
    msg = lib.cba_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(lib, rc: int, what: str) -> None:
    if rc != 0:
        raise BackendError(f"{what} failed (code {rc}): {last_error(lib)}")
