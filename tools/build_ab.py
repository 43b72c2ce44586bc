"""Build an A/B variant of the library: python tools/build_ab.py TAG [-DKNOB=1 ...]  ->  build_ab/libtolg_TAG.so

Same compile line as `_build.build_extension` plus the given flags, hazard lint included; the variants run through
TOLG_HIP_LIB (tools/ab_libs.sh).  build_ab/ is git-ignored and travels to the GPU box with the snapshot."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajectory_optimization_matrix_lie_groups_amd import _build, _dpp_lint  # noqa: E402


def main():
    tag, flags = sys.argv[1], sys.argv[2:]
    out = os.path.join(ROOT, "build_ab", "libtolg_%s.so" % tag)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", out, _build._SRC, *flags])
    isa = _dpp_lint.disassemble(out)
    findings = _dpp_lint.lint(isa) + _dpp_lint.lint_more(isa)
    for f in findings[:20]:
        print("HAZARD %s %x: %s" % f)
    if findings:
        os.replace(out, out + ".hazard")
        sys.exit(1)
    print(out)


if __name__ == "__main__":
    main()
