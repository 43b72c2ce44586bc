"""The C-ABI library loads and exports every symbol include/tolg.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import trajectory_optimization_matrix_lie_groups_amd as pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    so = pkg.build_extension()
    lib = ctypes.CDLL(so)
    hdr = open(os.path.join(ROOT, "include", "tolg.h")).read()
    declared = sorted(set(re.findall(r"\b(tolg_[a-z_]+)\s*\(", hdr)))
    assert len(declared) >= 9
    for name in declared:
        assert hasattr(lib, name), name
    from trajectory_optimization_matrix_lie_groups_amd import _capi
    assert sorted(_capi.SYMBOLS) == declared
    lib.tolg_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.tolg_version()


def test_workspace_query_and_argument_errors_without_gpu():
    from trajectory_optimization_matrix_lie_groups_amd import _capi
    lib = _capi.load()
    p = _capi.Problem()
    p.kind, p.m, p.N, p.dt = _capi.DYN_SE3, 6, 200, 0.05
    n = lib.tolg_workspace_bytes(ctypes.byref(p), 4096)
    # dominated by the linearisation scratch [N][12][B][13] doubles
    assert n > 200 * 12 * 4096 * 13 * 8
    p.m = 4  # SE3 dynamics has 6 inputs
    assert lib.tolg_workspace_bytes(ctypes.byref(p), 4096) == 0
    p.kind, p.m, p.dt = _capi.DYN_DRONE, 4, -1.0
    assert lib.tolg_workspace_bytes(ctypes.byref(p), 16) == 0


def test_solver_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from trajectory_optimization_matrix_lie_groups_amd import workloads, BatchedTrackingILQR
    prob, *_ = workloads.se3_tracking(4, N=20)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        BatchedTrackingILQR(prob, 4)
