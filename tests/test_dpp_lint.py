"""The DPP read-after-VALU-write lint (trajectory_optimization_matrix_lie_groups_amd/_dpp_lint.py): the rule on synthetic
disassembly, and the in-tree library clean (the backward sweep's inline-asm DPP blocks carry no blanket hazard nops)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajectory_optimization_matrix_lie_groups_amd import _build, _dpp_lint  # noqa: E402

HEAD = "0000000000001000 <k>:\n"
DPP = "\tv_fmac_f64_dpp v[10:11], v[4:5], v[6:7] row_newbcast:3 row_mask:0xf bank_mask:0xf// 000000001%03X: 0\n"


def _ins(text, addr):
    return "\t%s // 000000001%03X: 0\n" % (text, addr)


@pytest.mark.parametrize("between, n_findings", [
    ([], 1),                                            # writer directly in front
    (["v_mul_f64 v[20:21], v[22:23], v[24:25]"], 1),   # one wait state
    (["v_mul_f64 v[20:21], v[22:23], v[24:25]", "v_add_f64 v[30:31], v[22:23], v[24:25]"], 0),
    (["s_nop 0"], 1),
    (["s_nop 1"], 0),
    (["s_nop 0", "s_waitcnt lgkmcnt(0)"], 0),
])
def test_rule_on_synthetic_disassembly(between, n_findings):
    for writer in ("v_add_f64 v[4:5], v[0:1], v[2:3]", "v_accvgpr_read_b32 v5, a7"):
        t, a = HEAD + _ins(writer, 0), 8
        for b in between:
            t += _ins(b, a); a += 8
        t += DPP % a
        assert len(_dpp_lint.lint(t)) == n_findings, (writer, between)
    # a writer of an unrelated register, and the non-DPP operands, are no hazard
    t = HEAD + _ins("v_add_f64 v[6:7], v[0:1], v[2:3]", 0) + DPP % 8
    assert _dpp_lint.lint(t) == []


@pytest.mark.parametrize("between, n_findings", [
    (["s_mov_b32 s51, m0", "s_mov_b32 m0, s44", "s_nop 0"], 1),   # 3 wait states: what the LDS-DMA statement had
    (["s_mov_b32 s51, m0", "s_mov_b32 m0, s44", "s_nop 2"], 0),   # 5
    (["s_nop 3"], 1), (["s_nop 4"], 0),
])
def test_sgpr_written_by_valu_then_read_by_vector_memory(between, n_findings):
    t, a = HEAD + _ins("v_readfirstlane_b32 s64, v130", 0), 8
    for b in between:
        t += _ins(b, a); a += 8
    t += _ins("global_load_lds_dwordx4 v59, s[64:65] offset:1024", a)
    assert len(_dpp_lint.lint(t)) == n_findings
    # a scalar-ALU write of the base is no hazard
    t = HEAD + _ins("s_add_u32 s64, s64, s2", 0) + _ins("global_load_lds_dwordx4 v59, s[64:65]", 8)
    assert _dpp_lint.lint(t) == []


@pytest.mark.parametrize("seq, n_findings", [
    # 3: transcendental result -> non-transcendental VALU
    (["v_rcp_f64_e32 v[4:5], v[0:1]", "v_fma_f64 v[6:7], v[4:5], v[2:3], v[2:3]"], 1),
    (["v_rcp_f64_e32 v[4:5], v[0:1]", "s_nop 0", "v_fma_f64 v[6:7], v[4:5], v[2:3], v[2:3]"], 0),
    (["v_rcp_f64_e32 v[4:5], v[0:1]", "v_rsq_f64_e32 v[6:7], v[4:5]"], 0),
    # 4: SGPR / VCC written by a VALU instruction -> VALU read
    (["v_cmp_lt_f64_e64 s[4:5], v[0:1], v[2:3]", "v_cndmask_b32_e64 v7, v1, v2, s[4:5]"], 1),
    (["v_cmp_lt_f64_e64 s[4:5], v[0:1], v[2:3]", "v_mov_b32_e32 v9, v8", "v_cndmask_b32_e64 v7, v1, v2, s[4:5]"], 1),
    (["v_cmp_lt_f64_e64 s[4:5], v[0:1], v[2:3]", "s_nop 1", "v_cndmask_b32_e64 v7, v1, v2, s[4:5]"], 0),
    (["v_add_co_u32_e32 v1, vcc, v2, v3", "v_addc_co_u32_e32 v4, vcc, v5, v6, vcc"], 1),
    (["v_add_co_u32_e32 v1, vcc, v2, v3", "v_mov_b32_e32 v9, v8", "s_nop 0", "v_addc_co_u32_e32 v4, vcc, v5, v6, vcc"], 0),
    (["v_readfirstlane_b32 s7, v3", "v_mul_f64 v[4:5], s[6:7], v[0:1]"], 1),
    (["s_mov_b32 s7, s9", "v_mul_f64 v[4:5], s[6:7], v[0:1]"], 0),
    # (the SGPR pair of v_mad_u64_u32 / v_div_scale is an OUTPUT: a write behind a write is no hazard, a read behind it is)
    (["v_div_scale_f64 v[4:5], s[8:9], v[0:1], v[0:1], v[2:3]", "v_mad_u64_u32 v[6:7], s[8:9], s4, v1, v[2:3]"], 0),
    (["v_mad_u64_u32 v[6:7], s[8:9], s4, v1, v[2:3]", "v_cndmask_b32_e64 v7, v1, v2, s[8:9]"], 1),
    # 5: VGPR written by a VALU instruction -> readlane; lane select
    (["v_add_u32_e32 v3, v1, v2", "v_readfirstlane_b32 s7, v3"], 1),
    (["v_add_u32_e32 v3, v1, v2", "s_nop 0", "v_readfirstlane_b32 s7, v3"], 0),
    (["v_readfirstlane_b32 s7, v3", "s_nop 2", "v_readlane_b32 s8, v5, s7"], 1),
    (["v_readfirstlane_b32 s7, v3", "s_nop 3", "v_readlane_b32 s8, v5, s7"], 0),
    # 6: EXEC written by v_cmpx -> DPP
    (["v_cmpx_lt_f64_e32 v[0:1], v[2:3]", "s_nop 3", "v_mov_b32_dpp v4, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"], 1),
    (["v_cmpx_lt_f64_e32 v[0:1], v[2:3]", "s_nop 4", "v_mov_b32_dpp v4, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"], 0),
    # 7: M0 -> LDS-DMA
    (["s_mov_b32 m0, s31", "global_load_lds_dwordx4 v[28:29], off"], 1),
    (["s_mov_b32 m0, s31", "s_nop 0", "global_load_lds_dwordx4 v[28:29], off"], 0),
    (["s_mov_b32 m0, s31", "buffer_load_dword v1, s[4:7], 0 offen lds"], 1),
    (["s_mov_b32 m0, s31", "global_load_dwordx4 v[0:3], v[28:29], off"], 0),
])
def test_further_rules_on_synthetic_disassembly(seq, n_findings):
    t = HEAD
    for i, ins in enumerate(seq):
        t += _ins(ins, 8 * i)
    assert len(_dpp_lint.lint_more(t)) == n_findings, _dpp_lint.lint_more(t)


def _label(text, addr, target_off):
    return ("\t%s // 000000001%03X: 0 <k+0x%x>\n" % (text, addr, target_off))


def test_branch_into_the_window_is_followed_to_its_source():
    # the writer sits in front of a branch that jumps straight to the DPP read: the branch is the only wait state between them
    t = (HEAD + _ins("v_add_f64 v[4:5], v[0:1], v[2:3]", 0) + _label("s_cbranch_execz 4", 8, 0x20) + _ins("s_nop 7", 0x10)
         + _ins("s_nop 7", 0x18) + DPP % 0x20)
    f = _dpp_lint.lint(t)
    assert len(f) == 1 and "v_add_f64 at 1000" in f[0][2]
    # one more instruction behind the target: branch + instruction = the two wait states
    t = (HEAD + _ins("v_add_f64 v[4:5], v[0:1], v[2:3]", 0) + _label("s_cbranch_execz 4", 8, 0x20) + _ins("s_nop 7", 0x10)
         + _ins("s_nop 7", 0x18) + _ins("v_add_f64 v[30:31], v[0:1], v[2:3]", 0x20) + DPP % 0x28)
    assert _dpp_lint.lint(t) == []
    # a harmless instruction in front of the branch: both predecessors of the target are clean
    t = (HEAD + _ins("v_add_f64 v[8:9], v[0:1], v[2:3]", 0) + _label("s_cbranch_execz 4", 8, 0x20) + _ins("s_nop 7", 0x10)
         + _ins("s_nop 7", 0x18) + DPP % 0x20)
    assert _dpp_lint.lint(t) == []
    # an unconditional branch has no fall-through: the writer in front of it is not a predecessor of what follows
    t = (HEAD + _label("s_cbranch_scc1 3", 0, 0x18) + _ins("v_add_f64 v[4:5], v[0:1], v[2:3]", 8) + _label("s_branch 9", 0x10, 0x60)
         + DPP % 0x18)
    assert _dpp_lint.lint(t) == []


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") and not os.environ.get("ROCM_PATH"), reason="no llvm-objdump")
def test_in_tree_library_has_no_dpp_hazard():
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        _build.build_extension()
    lib = _build._SO
    assert os.path.exists(lib)
    text = _dpp_lint.disassemble(lib)
    assert text.count("_dpp") > 1000  # the blocks are there
    assert _dpp_lint.lint(text) == []
    assert _dpp_lint.lint_more(text) == []
