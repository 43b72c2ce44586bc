"""In-tree build of the HIP extension (hipcc cross-compiles gfx950 without a GPU)."""
import glob
import hashlib
import json
import os
import shutil
import subprocess
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "tolg_kernels.hip")
_SO = os.path.join(_HERE, "libtolg_hip.so")
_INFO = os.path.join(_HERE, "_build_info.json")


def _deps():
    """Every file the library is compiled from: the kernel source, the headers beside it, the C ABI header."""
    return [_SRC] + sorted(glob.glob(os.path.join(_HERE, "csrc", "*.h"))) + [os.path.join(os.path.dirname(_HERE), "include", "tolg.h")]


def lib_path():
    """The in-tree library; TOLG_HIP_LIB points at another build of the same ABI (A/B timing runs; bench.py
    refuses it unless asked, and reports it when it is used)."""
    return os.environ.get("TOLG_HIP_LIB") or _SO


def source_hash():
    """sha256 (first 16 hex digits) over the sources in `_deps()` as they are on disk now."""
    h = hashlib.sha256()
    for p in _deps():
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_info():
    """What `build_extension` recorded beside the library: source hash, git head, time (the GPU box gets the
    library and this file, not the .git directory)."""
    try:
        return json.load(open(_INFO))
    except Exception:
        return {}


def _stale():
    if not os.path.exists(_SO):
        return True
    t = os.path.getmtime(_SO)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in _deps())


def _git_head():
    try:
        root = os.path.dirname(_HERE)
        head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        dirty = subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "trajectory_optimization_matrix_lie_groups_amd/csrc",
                                         "include"], stderr=subprocess.DEVNULL).decode().strip()
        return head + ("+uncommitted-source-changes" if dirty else "")
    except Exception:
        return None


def build_extension(force=False, verbose=False, extra_flags=(), lint=True):
    """hipcc --offload-arch=gfx950 -> trajectory_optimization_matrix_lie_groups_amd/libtolg_hip.so"""
    if not force and not _stale():
        return _SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built (and there is no CPU fallback)")
    # built beside the target and moved into place when it has passed the lint: a process that loads the library while
    # another one rebuilds it (ranks of one job, pytest-xdist workers) sees the old file or the new one, never half of one
    tmp = "%s.%d.tmp" % (_SO, os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", tmp, _SRC, *extra_flags]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
        if lint:
            # the hand-written DPP blocks run without hazard nops where the emitted code keeps the distance: check that it does
            from . import _dpp_lint
            try:
                _isa = _dpp_lint.disassemble(tmp)
            except _dpp_lint.LintToolsMissing as e:
                # a usable library is not thrown away for want of binutils: loud, and the unit test of the lint will fail
                import warnings
                warnings.warn("HIP extension built WITHOUT the DPP hazard lint: %s" % e)
                _isa = ""
            findings = _dpp_lint.lint(_isa) + _dpp_lint.lint_more(_isa)
            if findings:
                bad = _SO + ".hazard"
                os.replace(tmp, bad)
                raise RuntimeError("DPP read-after-write hazards in the built library (kept as %s):\n%s"
                                   % (bad, "\n".join("%s %x: %s" % f for f in findings[:20])))
        os.replace(tmp, _SO)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    with open(_INFO, "w") as f:
        json.dump({"source_sha256_16": source_hash(), "git_head": _git_head(), "built_at": time.strftime("%Y-%m-%dT%H:%M:%S"),
                   "flags": list(extra_flags)}, f)
    return _SO
