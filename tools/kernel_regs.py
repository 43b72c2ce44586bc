"""Registers, LDS and scratch of the kernels in the built library (code-object metadata).
usage: python tools/kernel_regs.py [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_matrix_lie_groups_amd import _dpp_lint as L  # noqa: E402
import trajectory_optimization_matrix_lie_groups_amd as pkg  # noqa: E402


def main():
    pats = sys.argv[1:]
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([f"{L.llvm_dir()}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", pkg.lib_path(), os.path.join(d, "null")])
        subprocess.check_call([f"{L.llvm_dir()}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--targets={L.TARGET}", f"--output={co}"])
        notes = subprocess.check_output([f"{L.llvm_dir()}/llvm-readelf", "--notes", co]).decode()
    for e in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        e = ".agpr_count" + e
        m = re.search(r"\.name:\s+(\S+)", e)
        if not m:
            continue
        name = subprocess.check_output(["c++filt", m.group(1)]).decode().strip().split("(")[0]
        if pats and not any(p in name for p in pats):
            continue
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", e).group(1))  # noqa: E731
        print("%-62s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %5d" % (name[:62], g("vgpr_count"), g("agpr_count"), g("sgpr_count"),
                                                                     g("group_segment_fixed_size"), g("private_segment_fixed_size")))


if __name__ == "__main__":
    main()
