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


def test_integration_stub_structs_match_the_binding_and_the_header():
    """INTEGRATION.md's ctypes stub is what a maintainer pastes into the reference: its struct layouts must be the binding's
    (`_capi.Problem` / `_capi.Options`) and the header's, field for field (round 3 shipped a stub whose last `tolg_options`
    field was still called `reserved`)."""
    import re
    from trajectory_optimization_matrix_lie_groups_amd import _capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    hdr = open(os.path.join(root, "include", "tolg.h")).read()

    def stub_fields(cls):
        body = re.search(r"class %s\(C\.Structure\):.*?_fields_ = \[(.*?)\]\s*(?:#.*)?\n(?:class |lib)" % cls, md, re.S).group(1)
        out = []
        for name, typ, n in re.findall(r'\("(\w+)",\s*C\.(c_\w+)(?:\s*\*\s*(\d+))?\)', body):
            out.append((name, typ, int(n) if n else 1))
        return out

    def binding_fields(cls):
        out = []
        for name, typ in cls._fields_:
            n = getattr(typ, "_length_", 1)
            base = getattr(typ, "_type_", typ) if n > 1 else typ
            out.append((name, base.__name__, n))
        return out

    def header_fields(struct):
        end = hdr.index("} %s;" % struct)
        body = hdr[hdr.rindex("typedef struct {", 0, end) + len("typedef struct {"): end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for typ, name, n in re.findall(r"(int32_t|double)\s+(\w+)(?:\[(\d+)\])?\s*;", body):
            out.append((name, {"int32_t": "c_int", "double": "c_double"}[typ], int(n) if n else 1))
        return out

    norm = lambda fs: [(n, "c_int" if t in ("c_int", "c_int32") else t, k) for n, t, k in fs]  # noqa: E731
    for cls, struct in ((_capi.Problem, "tolg_problem"), (_capi.Options, "tolg_options")):
        name = cls.__name__
        assert norm(stub_fields(name)) == norm(binding_fields(cls)) == norm(header_fields(struct)), name
