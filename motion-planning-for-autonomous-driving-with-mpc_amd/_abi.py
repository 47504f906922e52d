"""
ctypes view of include/mpcgpu.h -- the C-ABI of libmpcgpu.so (hand-written HIP kernels for gfx950).

This is the only place the package touches the shared library.  There is NO CPU fallback: if the library is
missing or cannot be loaded, `load_library()` raises and every solver object fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libmpcgpu.so"
LIB_PATH = os.path.join(PKG_DIR, "csrc", LIB_NAME)

MPC_STATUS_CONVERGED = 1
MPC_STATUS_MAXITER = 0
MPC_STATUS_NAN = -6
MPC_STATUS_NOPROGRESS = -7

MPC_OK = 0
MPC_ERR_INVALID = -1
MPC_ERR_HIP = -2
MPC_ERR_BOUNDS = -3
MPC_ERR_STATE = -4

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class MpcProblemDesc(C.Structure):
    """mirror of `mpc_problem_desc` (include/mpcgpu.h)."""
    _fields_ = [
        ("N", C.c_int32), ("nx", C.c_int32), ("nu", C.c_int32), ("formulation", C.c_int32),
        ("max_iter", C.c_int32), ("fixed_iters", C.c_int32), ("obst_mult", C.c_int32), ("device", C.c_int32),
        ("dt", C.c_double), ("wheelbase", C.c_double), ("friction_div", C.c_double), ("ego_offset", C.c_double),
        ("tol", C.c_double),
        ("Q", C.c_double * 8), ("R", C.c_double * 2), ("P", C.c_double * 8), ("obstacle", C.c_double * 6),
    ]


EXPORTS = [
    "mpc_default_desc", "mpc_create", "mpc_destroy", "mpc_last_error", "mpc_set_bounds", "mpc_solve_batch",
    "mpc_solve_batch_dev", "mpc_plant_step", "mpc_set_profiling", "mpc_get_profile", "mpc_solve_batch_trace",
    "mpc_abi_version", "mpc_closed_loop_batch", "mpc_closed_loop_batch_dev", "mpc_metrics_batch", "mpc_forces_stage_eval", "mpc_forces_solve_batch",
    "mpc_get_pipeline_profile", "mpc_get_resident_profile", "mpc_measure_copy_bandwidth", "mpc_get_option", "mpc_plant_step_dev", "mpc_metrics_batch_dev", "mpc_forces_solve_batch_dev", "mpc_set_option", "mpc_last_rescued", "mpc_closed_loop_batch_ex", "mpc_closed_loop_batch_dev_ex", "mpc_last_loop_replayed", "mpc_validity_batch", "mpc_validity_batch_dev", "mpc_forces_closed_loop_batch", "mpc_forces_closed_loop_batch_dev",
]


class MpcLibraryError(RuntimeError):
    pass


_lib = None


def load_library(path: str | None = None):
    """dlopen libmpcgpu.so and declare the prototypes of include/mpcgpu.h; raises if it is not there."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if os.environ.get("MPCGPU_NO_TORCH") != "1":
        # torch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; a process must hold ONE HIP runtime
        # (and stream handles passed to mpc_solve_batch_dev must belong to it), so let torch load its copy first
        # and libmpcgpu.so binds to the same SONAME.  Loading in the other order leaves torch without a GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(path):
        raise MpcLibraryError(
            f"{path} not found: the HIP extension is not built. Run `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:
        L = C.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise MpcLibraryError(f"cannot load {path}: {e}") from e
    vp = C.c_void_p
    L.mpc_default_desc.argtypes = [C.POINTER(MpcProblemDesc), C.c_int32, C.c_int32]
    L.mpc_default_desc.restype = None
    L.mpc_create.argtypes = [C.POINTER(vp), C.POINTER(MpcProblemDesc)]
    L.mpc_create.restype = C.c_int
    L.mpc_destroy.argtypes = [vp]
    L.mpc_destroy.restype = C.c_int
    L.mpc_last_error.argtypes = [vp]
    L.mpc_last_error.restype = C.c_char_p
    L.mpc_set_bounds.argtypes = [vp, _dp, _dp, _dp, _dp]
    L.mpc_set_bounds.restype = C.c_int
    L.mpc_solve_batch.argtypes = [vp, C.c_int32, _dp, _dp, _dp, _dp, _ip, _ip, _dp]
    L.mpc_solve_batch.restype = C.c_int
    L.mpc_solve_batch_dev.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_solve_batch_dev.restype = C.c_int
    L.mpc_plant_step.argtypes = [vp, C.c_int32, C.c_int32, _dp, _dp, _dp]
    L.mpc_plant_step.restype = C.c_int
    L.mpc_set_profiling.argtypes = [vp, C.c_int32]
    L.mpc_set_profiling.restype = C.c_int
    L.mpc_get_profile.argtypes = [vp, _dp]
    L.mpc_get_profile.restype = C.c_int
    L.mpc_get_pipeline_profile.argtypes = [vp, _dp]
    L.mpc_get_pipeline_profile.restype = C.c_int
    L.mpc_get_resident_profile.argtypes = [vp, _dp]
    L.mpc_get_resident_profile.restype = C.c_int
    L.mpc_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.mpc_get_option.restype = C.c_int
    L.mpc_measure_copy_bandwidth.argtypes = [vp, C.c_size_t, C.c_int32, _dp]
    L.mpc_measure_copy_bandwidth.restype = C.c_int
    L.mpc_solve_batch_trace.argtypes = [vp, C.c_int32, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp, C.c_int32, _ip]
    L.mpc_solve_batch_trace.restype = C.c_int
    L.mpc_closed_loop_batch.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp, _ip]
    L.mpc_closed_loop_batch.restype = C.c_int
    L.mpc_closed_loop_batch_dev.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_closed_loop_batch_dev.restype = C.c_int
    L.mpc_metrics_batch.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp, C.c_double, C.c_int32, _dp, _dp, _dp]
    L.mpc_metrics_batch.restype = C.c_int
    L.mpc_forces_stage_eval.argtypes = [vp, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    L.mpc_forces_stage_eval.restype = C.c_int
    L.mpc_forces_solve_batch.argtypes = [vp, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int32, _dp, _ip, _ip, _dp]
    L.mpc_forces_solve_batch.restype = C.c_int
    L.mpc_plant_step_dev.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp]
    L.mpc_plant_step_dev.restype = C.c_int
    L.mpc_metrics_batch_dev.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, C.c_double, C.c_int32, vp, vp, vp, vp]
    L.mpc_metrics_batch_dev.restype = C.c_int
    L.mpc_forces_solve_batch_dev.argtypes = [vp, C.c_int32, vp, vp, vp, _dp, _dp, _dp, _dp, C.c_int32, vp, vp, vp, vp, vp]
    L.mpc_forces_solve_batch_dev.restype = C.c_int
    L.mpc_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.mpc_set_option.restype = C.c_int
    L.mpc_closed_loop_batch_ex.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp, C.c_int32, C.c_double, C.c_uint64, _dp, _dp, _ip]
    L.mpc_closed_loop_batch_ex.restype = C.c_int
    L.mpc_closed_loop_batch_dev_ex.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_int32, C.c_double, C.c_uint64, vp, vp, vp, vp]
    L.mpc_closed_loop_batch_dev_ex.restype = C.c_int
    L.mpc_last_loop_replayed.argtypes = [vp]
    L.mpc_last_loop_replayed.restype = C.c_int
    L.mpc_validity_batch.argtypes = [vp, C.c_int32, C.c_int32, _dp, C.c_double, C.c_double, C.c_int32, _dp, C.c_int32, _dp, C.c_int32, _dp, _ip, _ip]
    L.mpc_validity_batch.restype = C.c_int
    L.mpc_validity_batch_dev.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_double, C.c_double, C.c_int32, vp, C.c_int32, vp, C.c_int32, vp, vp, vp, vp]
    L.mpc_validity_batch_dev.restype = C.c_int
    L.mpc_forces_closed_loop_batch.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int32, C.c_int32,
                                               C.c_double, C.c_uint64, _dp, _dp, _ip]
    L.mpc_forces_closed_loop_batch.restype = C.c_int
    L.mpc_forces_closed_loop_batch_dev.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, _dp, _dp, _dp, _dp, C.c_int32, C.c_int32,
                                                   C.c_double, C.c_uint64, vp, vp, vp, vp]
    L.mpc_forces_closed_loop_batch_dev.restype = C.c_int
    L.mpc_last_rescued.argtypes = [vp]
    L.mpc_last_rescued.restype = C.c_int
    L.mpc_abi_version.argtypes = []
    L.mpc_abi_version.restype = C.c_int
    if path == LIB_PATH:
        _lib = L
    return L


def as_dp(a):
    return None if a is None else a.ctypes.data_as(_dp)


def as_ip(a):
    return None if a is None else a.ctypes.data_as(_ip)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a
