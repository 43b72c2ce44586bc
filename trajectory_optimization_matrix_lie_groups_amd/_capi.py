"""ctypes binding of include/tolg.h (the drop-in boundary).  Plain pointers and sizes only."""
import ctypes as C
import os

from ._build import lib_path

DYN_SE3, DYN_RIGIDBODY, DYN_DRONE, DYN_SO3, DYN_PENDULUM3D = 0, 1, 2, 3, 4
MODE_MS, MODE_SS = 0, 1
ST_OK, ST_MAXREG, ST_NODESCENT, ST_NONFINITE, ST_INTERNAL = 0, 1, 2, 3, 4
SCHED_AUTO, SCHED_SPLIT = 0, 1
_ERR = {-1: "bad argument", -2: "workspace too small", -3: "kernel launch failed", -4: "inertia matrix singular"}


class Problem(C.Structure):
    _fields_ = [("kind", C.c_int32), ("m", C.c_int32), ("N", C.c_int32), ("reserved", C.c_int32),
                ("dt", C.c_double), ("J", C.c_double * 36), ("Q", C.c_double * 144), ("P", C.c_double * 144),
                ("R", C.c_double * 36), ("pend_mass", C.c_double), ("pend_length", C.c_double)]


class Options(C.Structure):
    _fields_ = [("mode", C.c_int32), ("max_iter", C.c_int32), ("line_search", C.c_int32),
                ("rollout_linear", C.c_int32), ("tol_grad", C.c_double), ("tol_defect", C.c_double),
                ("max_reg", C.c_double), ("schedule", C.c_int32), ("check_every", C.c_int32)]


_lib = None
SYMBOLS = ["tolg_workspace_bytes", "tolg_create", "tolg_destroy", "tolg_solve_batch", "tolg_solve_begin",
           "tolg_solve_iterate", "tolg_solve_iterate_until", "tolg_solve_end", "tolg_solve_peek", "tolg_solve_active_count", "tolg_set_al", "tolg_al_update", "tolg_eval_knot", "tolg_linearize_backward",
           "tolg_rollout", "tolg_expected_change", "tolg_kernel_time", "tolg_enable_timing", "tolg_version", "tolg_selftest_series"]


def load():
    """Load libtolg_hip.so or raise: the product has no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    so = lib_path()
    if not os.path.exists(so):
        raise RuntimeError(
            "HIP extension %s is missing. Build it with "
            "trajectory_optimization_matrix_lie_groups_amd.build_extension() (needs hipcc); "
            "there is no CPU fallback." % so)
    lib = C.CDLL(so)
    vp, dp, ip = C.c_void_p, C.c_void_p, C.c_void_p
    lib.tolg_workspace_bytes.restype = C.c_size_t
    lib.tolg_workspace_bytes.argtypes = [C.POINTER(Problem), C.c_int32]
    lib.tolg_create.restype = C.c_int
    lib.tolg_create.argtypes = [C.POINTER(Problem), dp, dp, C.c_int32, vp, C.c_size_t, vp, C.POINTER(vp)]
    lib.tolg_destroy.restype = None
    lib.tolg_destroy.argtypes = [vp]
    lib.tolg_solve_batch.restype = C.c_int
    lib.tolg_solve_batch.argtypes = [vp, C.POINTER(Options), C.c_int32] + [dp] * 11 + [ip] * 3 + [vp]
    lib.tolg_solve_begin.restype = C.c_int
    lib.tolg_solve_begin.argtypes = [vp, C.POINTER(Options), C.c_int32] + [dp] * 8 + [vp]
    lib.tolg_solve_iterate.restype = C.c_int
    lib.tolg_solve_iterate.argtypes = [vp, C.c_int32, vp]
    lib.tolg_solve_iterate_until.restype = C.c_int
    lib.tolg_solve_iterate_until.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_int32), vp]
    lib.tolg_solve_end.restype = C.c_int
    lib.tolg_solve_end.argtypes = [vp, dp, dp, dp, ip, ip, ip, vp]
    lib.tolg_solve_peek.restype = C.c_int
    lib.tolg_solve_peek.argtypes = [vp, dp, dp, dp, ip, ip, ip, vp]
    lib.tolg_solve_active_count.restype = C.c_int
    lib.tolg_solve_active_count.argtypes = [vp, ip, vp]
    lib.tolg_set_al.restype = C.c_int
    lib.tolg_set_al.argtypes = [vp, dp, dp, dp, dp]
    lib.tolg_al_update.restype = C.c_int
    lib.tolg_al_update.argtypes = [vp, C.c_int32, dp, dp, dp, dp, dp, dp, C.c_double, C.c_double, C.c_double, dp, ip, vp]
    lib.tolg_eval_knot.restype = C.c_int
    lib.tolg_eval_knot.argtypes = [vp, C.c_int32, C.c_int32] + [dp] * 13 + [vp]
    lib.tolg_linearize_backward.restype = C.c_int
    lib.tolg_linearize_backward.argtypes = [vp, C.c_int32, C.c_double, C.c_int32] + [dp] * 13 + [vp]
    lib.tolg_rollout.restype = C.c_int
    lib.tolg_rollout.argtypes = [vp, C.c_int32, C.c_int32, C.c_double, C.c_int32, dp, dp, dp, vp]
    lib.tolg_expected_change.restype = C.c_int
    lib.tolg_expected_change.argtypes = [vp, C.c_int32, C.c_int32, dp, ip, vp]
    lib.tolg_kernel_time.restype = C.c_int
    lib.tolg_kernel_time.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.tolg_enable_timing.restype = None
    lib.tolg_enable_timing.argtypes = [vp, C.c_int32]
    lib.tolg_version.restype = C.c_char_p
    lib.tolg_version.argtypes = []
    lib.tolg_selftest_series.restype = C.c_int
    lib.tolg_selftest_series.argtypes = [C.c_int32, dp, dp, vp]
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (rc=%d)" % (what, _ERR.get(rc, "unknown"), rc))
