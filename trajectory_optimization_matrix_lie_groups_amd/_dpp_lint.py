#!/usr/bin/env python3
"""DPP read-after-VALU-write lint over the built library's gfx950 code.

The backward sweep's products are hand-written `v_fmac_f64_dpp` / `v_mov_b64_dpp` blocks (inline asm).  CDNA has no
interlock for "VALU writes a VGPR, a DPP instruction reads it as its DPP operand (src0) within the next two wait
states": the read returns the old value.  The compiler pads its own DPP instructions; it does not look into inline
asm, and it is free to schedule the instruction that produces an asm operand (an add, a copy, a v_accvgpr_read)
directly in front of the block.  Most blocks here run without a leading s_nop (the sweep is issue-bound and an
s_nop costs as much as a multiply-add, profiles/r03_valu_issue_microbench.txt), which is only sound if the code the
compiler actually emitted keeps the distance -- so this lint checks exactly that, on every build:

    for every DPP instruction: none of the instructions in the two wait states in front of it (program order, s_nop N
    counting N + 1) is a VALU instruction that writes a register of the DPP operand.

Second rule, same reason (the LDS-DMA bursts are inline asm with a scalar base address): a VALU instruction that writes
an SGPR (v_readfirstlane, v_readlane, a carry-out, a compare) must be five wait states ahead of a vector-memory
instruction that reads that SGPR.

Rules three to seven (late round 3) are the rest of what LLVM's hazard recogniser pads on gfx940-class parts and cannot
see through an inline-asm statement -- checked over ALL code, so the compiler's own padding is verified along the way:
  3. a transcendental result (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos) read by a non-transcendental
     VALU instruction: 1 wait state;
  4. an SGPR or VCC written by a VALU instruction (compare, carry-out, readlane) read by a VALU instruction: 2;
  5. a VGPR written by a VALU instruction read by v_readlane / v_readfirstlane / v_writelane: 1; their lane select
     from a VALU-written SGPR: 4;
  6. EXEC written by a VALU instruction (v_cmpx) in front of a DPP instruction: 5;
  7. M0 written by a scalar instruction in front of an LDS-DMA load (`global_load_lds_*`, `buffer_load_* ... lds`): 1.

Rule 1 follows control flow: at a branch target inside the window the scan continues along the fall-through predecessor
and from every branch that jumps there (an earlier version reported the target itself, which refused builds whose other
predecessor was harmless).

    python tools/dpp_hazard_lint.py [path/to/libtolg_hip.so]      exit code 1 on a finding

`_build.build_extension` runs it on every build and refuses a library with findings; tests/test_dpp_lint.py runs it on
the in-tree library.
"""
import os
import re
import subprocess
import sys
import tempfile

TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


class LintToolsMissing(RuntimeError):
    """llvm-objcopy / clang-offload-bundler / llvm-objdump were not found beside the ROCm install."""


def llvm_dir():
    """Directory of the LLVM binutils of the ROCm install in use: beside the hipcc on PATH, then ROCM_PATH, then /opt/rocm."""
    import shutil
    cands = []
    hipcc = shutil.which("hipcc")
    if hipcc:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root:
            cands.append(os.path.join(root, "lib", "llvm", "bin"))
    for c in cands:
        if all(os.path.exists(os.path.join(c, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")):
            return c
    raise LintToolsMissing("the hazard lint needs llvm-objcopy, clang-offload-bundler and llvm-objdump (looked in %s); "
                           "build_extension(lint=False) builds without it" % ", ".join(cands))


def disassemble(lib):
    LLVM = llvm_dir()
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(d, "null")])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}",
                               f"--output={co}"])
        return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co]).decode()


_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def _sregs(tok):
    out = set()
    for m in _SREG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _is_vmem(op):
    return op.startswith(("global_", "buffer_", "flat_", "scratch_", "tbuffer_"))


def _regs(tok):
    out = set()
    for m in _REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _is_valu(op):
    return op.startswith("v_") and not op.startswith("v_nop")


def _is_dpp(op, args):
    return op.endswith("_dpp") or any(k in args for k in ("row_newbcast", "row_shl", "row_shr", "quad_perm", "row_bcast", "row_ror",
                                                            "row_mirror", "wave_", "row_half_mirror"))


def _parse(text):
    """-> {kernel: [(address, opcode, args)]}, {kernel: {target address: [indices of the branches that jump there]}}"""
    kernels, order, base = {}, [], {}
    kernel = None
    raw_targets = []
    for ln in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", ln)
        if m:
            kernel = m.group(2)
            kernels[kernel] = []
            base[kernel] = int(m.group(1), 16)
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if not m or kernel is None:
            continue
        op, args, addr = m.group(1), m.group(2), int(m.group(3), 16)
        kernels[kernel].append((addr, op, args))
        t = re.search(r"<([^>+]+)\+0x([0-9a-f]+)>", ln) if re.match(r"s_c?branch", op) else None
        if t:
            raw_targets.append((kernel, len(kernels[kernel]) - 1, t.group(1), int(t.group(2), 16)))
    sources = {k: {} for k in kernels}
    for k, idx, sym, off in raw_targets:
        if sym in base and sym == k:
            sources[k].setdefault(base[k] + off, []).append(idx)
    return kernels, sources


def lint(text):
    """Rules 1 and 2 -> list of (kernel, address, message).  Rule 1 follows control flow: at a branch target the scan goes
    on along the fall-through predecessor AND from every branch that jumps there (a branch is one wait state itself)."""
    findings = []
    kernels, sources = _parse(text)
    for kernel, ins in kernels.items():
        src = sources[kernel]

        def vwrites(op, args):
            if not _is_valu(op) or op.startswith("v_cmp") or op.startswith("v_accvgpr_write"):
                return set()
            toks = [t.strip() for t in args.split(",")]
            return _regs(toks[0]) if toks else set()

        def walk(i, ws, src0, op_d, addr_d, depth, seen):
            """instructions in front of index i (exclusive), `ws` wait states already between them and the DPP read"""
            k = i - 1
            while ws < 2 and k >= 0:
                a, op, args = ins[k]
                # is the instruction BEHIND k (index k + 1) a branch target?  then every branch to it is a predecessor too
                if ins[k + 1][0] in src and depth < 4:
                    for s in src[ins[k + 1][0]]:
                        if (s, ws) not in seen:
                            seen.add((s, ws))
                            walk(s + 1, ws, src0, op_d, addr_d, depth + 1, seen)
                if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                    return  # no fall-through from here
                w = vwrites(op, args)
                if w & src0:
                    findings.append((kernel, addr_d, "%s reads v%s through DPP %d wait state(s) after %s at %x wrote it"
                                     % (op_d, sorted(w & src0), ws, op, a)))
                ws += (int(args.split()[0], 0) + 1) if (op == "s_nop" and args) else 1
                k -= 1
            if ws < 2 and k < 0 and depth == 0:
                pass  # kernel entry: nothing in front

        swindow = []
        for i, (addr, op, args) in enumerate(ins):
            if _is_dpp(op, args):
                toks = [t.strip() for t in args.split(",")]
                src0 = _regs(toks[1].split()[0]) if len(toks) > 1 else set()  # src0 is the DPP operand
                walk(i, 0, src0, op, addr, 0, set())
            if _is_vmem(op):
                sread = _sregs(args)
                ws = 0
                for w_ws, w_op, w_regs, w_addr in reversed(swindow):
                    if ws >= 5:
                        break
                    if w_regs & sread:
                        findings.append((kernel, addr, "%s reads s%s %d wait state(s) after %s at %x wrote it (VALU write of an SGPR -> "
                                         "vector memory: 5)" % (op, sorted(w_regs & sread), ws, w_op, w_addr)))
                    ws += w_ws
            if op == "s_nop":
                swindow.append(((int(args.split()[0], 0) if args else 0) + 1, op, set(), addr))
            else:
                toks = [t.strip() for t in args.split(",")]
                swritten = set()
                if _is_valu(op) and toks:
                    if op.startswith(("v_readfirstlane", "v_readlane")) or "_co_" in op or op.startswith(("v_cmp", "v_div_scale")):
                        swritten = _sregs(toks[0]) | (_sregs(toks[1]) if ("_co_" in op or op.startswith("v_div_scale")) and len(toks) > 1 else set())
                swindow.append((1, op, swritten, addr))
            swindow = swindow[-12:]
    # one finding per (DPP instruction, writer): paths that merge report the same pair twice
    out, seen_f = [], set()
    for f in findings:
        if f not in seen_f:
            seen_f.add(f)
            out.append(f)
    return out


_TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
_DPP_CTRL = ("row_newbcast", "row_shl", "row_shr", "quad_perm", "row_bcast", "row_ror", "row_mirror", "wave_", "row_half_mirror")


def _carry_form(op):
    # the second operand is an SGPR-pair OUTPUT (carry / scale flag), not a source
    return "_co_" in op or op.startswith(("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32"))


def lint_more(text):
    """Rules 3-7 of the module docstring -> list of (kernel, address, message)."""
    findings, kernel, prev = [], None, []   # prev: (wait states, opcode, VGPRs written, SGPRs written, exec written, m0 written)
    for ln in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", ln)
        if m:
            kernel, prev = m.group(2), []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if not m or kernel is None:
            continue
        op, args, addr = m.group(1), m.group(2), int(m.group(3), 16)
        toks = [t.strip() for t in args.split(",")]
        if _is_valu(op):
            srcs = ",".join(toks[2:] if _carry_form(op) else toks[1:])
            vread, sread = _regs(srcs), set(_sregs(srcs))
            if op.startswith("v_cmp") and not op.startswith("v_cmpx") and toks and toks[0].startswith("v"):
                vread |= _regs(toks[0])        # e32 compare: the first token is a source
            if "vcc" in srcs:
                sread.add("vcc")
            is_dpp = op.endswith("_dpp") or any(k in args for k in _DPP_CTRL)
            lane_op = op.startswith(("v_readlane", "v_readfirstlane", "v_writelane"))
            ws = 0
            for w_ws, w_op, vw, sw, ew, _m0 in reversed(prev):
                if ws < 1 and w_op.startswith(_TRANS) and not op.startswith(_TRANS) and (vw & vread):
                    findings.append((kernel, addr, "%s reads v%s %d wait state(s) after the transcendental %s wrote it (1)"
                                     % (op, sorted(vw & vread), ws, w_op)))
                if ws < 2 and (sw & sread):
                    findings.append((kernel, addr, "%s reads %s %d wait state(s) after the VALU instruction %s wrote it (2)"
                                     % (op, sorted(map(str, sw & sread)), ws, w_op)))
                if ws < 1 and lane_op and (vw & vread):
                    findings.append((kernel, addr, "%s reads v%s %d wait state(s) after %s wrote it (1)" % (op, sorted(vw & vread), ws, w_op)))
                if ws < 4 and op.startswith(("v_readlane", "v_writelane")) and len(toks) > 2 and (sw & set(_sregs(toks[2]))):
                    findings.append((kernel, addr, "%s takes its lane select %d wait state(s) after %s wrote it (4)" % (op, ws, w_op)))
                if ws < 5 and ew and is_dpp:
                    findings.append((kernel, addr, "DPP %s %d wait state(s) after %s wrote EXEC (5)" % (op, ws, w_op)))
                ws += w_ws
                if ws >= 5:
                    break
        if _is_vmem(op) and ("_lds_" in op or re.search(r"\blds\b", args)):
            if prev and prev[-1][5]:
                findings.append((kernel, addr, "%s directly behind %s (M0 write -> LDS-DMA: 1)" % (op, prev[-1][1])))
        if op == "s_nop":
            prev.append(((int(args.split()[0], 0) if args else 0) + 1, op, set(), set(), False, False))
        else:
            vw, sw, ew, m0w = set(), set(), False, False
            if _is_valu(op) and toks:
                if op.startswith("v_cmpx"):
                    ew = True
                elif op.startswith("v_cmp"):
                    sw = set(_sregs(toks[0])) | ({"vcc"} if (toks[0] == "vcc" or toks[0].startswith("v")) else set())
                elif op.startswith(("v_readfirstlane", "v_readlane")):
                    sw = set(_sregs(toks[0]))
                elif _carry_form(op):
                    vw = _regs(toks[0])
                    sw = set(_sregs(toks[1])) | ({"vcc"} if len(toks) > 1 and "vcc" in toks[1] else set())
                elif not op.startswith("v_accvgpr_write"):
                    vw = _regs(toks[0])
            elif op.startswith("s_") and toks and toks[0] == "m0":
                m0w = True
            prev.append((1, op, vw, sw, ew, m0w))
        prev = prev[-12:]
    return findings


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtolg_hip.so")
    text = disassemble(lib)
    f = lint(text) + lint_more(text)
    for k, a, msg in f:
        print("%s  %x: %s" % (k, a, msg))
    print("%d finding(s) in %s" % (len(f), lib))
    return 1 if f else 0


if __name__ == "__main__":
    sys.exit(main())
