"""In-tree build of the HIP extension (hipcc cross-compiles gfx950 without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "tolg_kernels.hip")
_DEPS = [_SRC, os.path.join(_HERE, "csrc", "tolg_lie.h"), os.path.join(os.path.dirname(_HERE), "include", "tolg.h")]
_SO = os.path.join(_HERE, "libtolg_hip.so")


def lib_path():
    """The in-tree library; TOLG_HIP_LIB points at another build of the same ABI (A/B timing runs)."""
    return os.environ.get("TOLG_HIP_LIB") or _SO


def _stale():
    if not os.path.exists(_SO):
        return True
    t = os.path.getmtime(_SO)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in _DEPS)


def build_extension(force=False, verbose=False, extra_flags=()):
    """hipcc --offload-arch=gfx950 -> trajectory_optimization_matrix_lie_groups_amd/libtolg_hip.so"""
    if not force and not _stale():
        return _SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built (and there is no CPU fallback)")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", _SO, _SRC, *extra_flags]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _SO
