#!/usr/bin/env python3
"""Writes trajectory_optimization_matrix_lie_groups_amd/csrc/tolg_k2_blocks.h: the interleaved inline-asm blocks of
k_backward3 (tolg_backward3.h) for m = 6.

Why machine-written: the three latency chains of a Riccati step -- the L Dl L^T factorisation (a reciprocal and two
Newton steps per pivot), the forward and the back substitution -- each leave about half of the issue slots of the
one wavefront a SIMD holds empty (a dependent fp64 instruction issues every ~9 cycles, an independent one every
~5), while the three products that surround them (Q_xx = l_xx + F_x^T Z, the symmetrisation, the rank-m update) are
pure issue.  The compiler cannot interleave them: the DPP-fused multiply-adds only exist as inline asm.  So each
chain step is emitted as one asm statement with its share of the neighbouring product's instructions placed in the
gaps of the chain.  Hazards the assembler does not see (cdna4 ISA, manually inserted wait states): a VGPR written by
a VALU instruction needs two wait states before a DPP instruction reads it as its DPP operand; the result of a
transcendental (v_rcp_f64) one wait state before a non-transcendental VALU instruction reads it -- both are kept by
construction (fillers in between) and checked by `hazard_check` below.  A statement writes its outputs long before it
has read all of its inputs: write-only outputs are early-clobber ("=&v"), and a read-write operand ("+v") must never
enter a statement holding a copy of one of the statement's inputs -- the compiler then gives both one register (it
did, for t = copy of Y: the back substitution therefore accumulates from zero and adds y through zn = -y / Dl).

Blocks (M = 6; lane of pivot / input u is 6 + u):
  b3_piv<GRAV, J>   pivot J of the factorisation + chunk J of Qh += F_x^T Z
  b3_fwd<K>         column K of the forward substitution + the symmetrisation of Qh[2K], Qh[2K+1]
  b3_bwd<K>         column K of the back substitution + rank-1 update K' of V = Qh + Y^T zn
"""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "trajectory_optimization_matrix_lie_groups_amd",
                   "csrc", "tolg_k2_blocks.h")
M = 6


def lane(u):
    return 6 + u


class Block:
    """Collects operands (C++ expression -> %n) and instruction lines of one asm statement."""

    def __init__(self):
        self.outs, self.ins, self.lines = [], [], []

    def _find(self, lst, expr):
        for i, (e, _) in enumerate(lst):
            if e == expr:
                return i
        return None

    def out(self, expr, cons="+v"):
        if self._find(self.outs, expr) is None:
            self.outs.append((expr, cons))
        return ("O", expr)

    def inp(self, expr, cons="v"):
        if self._find(self.outs, expr) is not None:
            return ("O", expr)
        if self._find(self.ins, expr) is None:
            self.ins.append((expr, cons))
        return ("I", expr)

    def ref(self, h):
        kind, expr = h
        if kind == "O":
            return "%%%d" % self._find(self.outs, expr)
        return "%%%d" % (len(self.outs) + self._find(self.ins, expr))

    def emit(self, fmt, *hs, writes=None, dpp_src=None, reads=(), trans=False):
        self.lines.append(dict(fmt=fmt, hs=hs, writes=writes, dpp_src=dpp_src, reads=reads, trans=trans))

    def render(self, indent="  "):
        assert len(self.outs) + len(self.ins) <= 30, "asm operand limit"
        hazard_check(self.lines)
        txt = []
        for ln in self.lines:
            txt.append('"' + ln["fmt"].format(*[self.ref(h) for h in ln["hs"]]) + '\\n\\t"')
        body = ("\n" + indent + "             ").join(txt)
        outs = ", ".join('"%s"(%s)' % (c, e) for e, c in self.outs)
        ins = ", ".join('"%s"(%s)' % (c, e) for e, c in self.ins)
        return indent + "asm volatile(" + body + "\n" + indent + "             : " + outs + "\n" + indent + "             : " + ins + ");\n"


def hazard_check(lines):
    """VALU write -> DPP read of the same register as DPP operand: >= 2 instructions in between;
    trans write -> VALU read: >= 1 instruction in between (every instruction here is a VALU instruction)."""
    for i, ln in enumerate(lines):
        for back in (1, 2):
            if i - back < 0:
                continue
            p = lines[i - back]
            if p["writes"] is not None and ln["dpp_src"] is not None and p["writes"] == ln["dpp_src"]:
                raise AssertionError("DPP hazard: %r then %r" % (p["fmt"], ln["fmt"]))
        if i >= 1:
            p = lines[i - 1]
            if p.get("trans") and p["writes"] is not None and (p["writes"] in ln["reads"] or p["writes"] == ln["dpp_src"]):
                raise AssertionError("trans hazard: %r then %r" % (p["fmt"], ln["fmt"]))


def interleave(chain, fill, lead=0, per_gap=1):
    """chain ops in order with `per_gap` fillers after each (more after the ops listed in chain[i]['gap']);
    `lead` fillers first; leftover fillers at the end."""
    out, f = [], list(fill)
    for _ in range(lead):
        if f:
            out.append(f.pop(0))
    for c in chain:
        out.append(c)
        for _ in range(c.get("gap", per_gap)):
            if f:
                out.append(f.pop(0))
    out.extend(f)
    return out


DPPF = "v_fmac_f64_dpp {0}, {1}, {2} row_newbcast:%d row_mask:0xf bank_mask:0xf"


def qh_chunk(b, grav, ks):
    """Qh[r] += A[k]@lane r * Z[k] for the structurally non-zero rows r of F_x's row k (one entry per FMA)."""
    ops = []
    for k in ks:
        if k < 3:
            rows = [0, 1, 2, 6, 7, 8]
        elif k < 6:
            rows = list(range(12))
        else:
            rows = ([0, 1, 2] if grav else []) + list(range(6, 12))
        for r in rows:
            acc = b.out("Qh[%d]" % r)
            a = b.inp("A[%d]" % k)
            z = b.inp("Z[%d]" % k)
            ops.append(dict(fmt=DPPF % r, hs=(acc, a, z), writes=acc, dpp_src=a, reads=(z, acc)))
    return ops


QH_CHUNKS = [[0, 1, 2], [3], [4], [5], [6, 7, 8], [9, 10, 11]]


def gen_piv(grav, J):
    b = Block()
    R = M - 1 - J
    aJ = b.inp("a[%d]" % J)
    rows = [b.out("a[%d]" % i) for i in range(J + 1, M)]
    r = b.out("rinv", "=&v")
    d = b.out("d", "=&v")
    e = b.out("e", "=&v")
    chain = []
    if R > 0:
        wm = b.inp("wm")
        pre = b.out("pre", "=&v")
        w = b.out("w", "=&v")
        chain.append(dict(fmt="v_mul_f64 {0}, {1}, {2}", hs=(pre, aJ, wm), writes=pre, dpp_src=None, reads=(aJ, wm), gap=2))
    chain.append(dict(fmt="v_mov_b64_dpp {0}, {1} row_newbcast:%d row_mask:0xf bank_mask:0xf" % lane(J), hs=(d, aJ), writes=d,
                      dpp_src=aJ, reads=(), gap=1))
    chain.append(dict(fmt="v_rcp_f64 {0}, {1}", hs=(r, d), writes=r, dpp_src=None, reads=(d,), trans=True, gap=2))
    for _ in range(2):
        chain.append(dict(fmt="v_fma_f64 {0}, -{1}, {2}, 1.0", hs=(e, d, r), writes=e, dpp_src=None, reads=(d, r), gap=1))
        chain.append(dict(fmt="v_fma_f64 {0}, {0}, {1}, {0}", hs=(r, e), writes=r, dpp_src=None, reads=(r, e), gap=1))
    if R > 0:
        chain.append(dict(fmt="v_mul_f64 {0}, {1}, {2}", hs=(w, pre, r), writes=w, dpp_src=None, reads=(pre, r), gap=1))
        for ai in rows:
            chain.append(dict(fmt=DPPF % lane(J), hs=(ai, ai, w), writes=ai, dpp_src=ai, reads=(w, ai), gap=1))
    fill = qh_chunk(b, grav, QH_CHUNKS[J])
    # J = 0: a[0] was written by the compiler's code right in front of the statement: keep the DPP read two slots away
    b.lines = interleave(chain, fill, lead=(0 if R > 0 else 2))
    if R == 0:
        pass
    name = "b3_piv"
    sig = ("template <>\nTOLG_DEV void %s<%s, %d>(double (&a)[6], double& rinv, double& d, double wm, double (&Qh)[12], "
           "const double (&A)[12], const double (&Z)[12]) {\n" % (name, "true" if grav else "false", J))
    tmp = "  double e" + (", pre, w" if R > 0 else "") + ";\n"
    return sig + tmp + b.render() + "}\n"


def gen_fwd(K):
    b = Block()
    yK = b.inp("y[%d]" % K)
    nri = b.inp("nri")
    zn = b.out("zn", "=&v")
    chain = [dict(fmt="v_mul_f64 {0}, {1}, {2}", hs=(zn, yK, nri), writes=zn, dpp_src=None, reads=(yK, nri), gap=2)]
    for i in range(K + 1, M):
        yi = b.out("y[%d]" % i)
        ai = b.inp("a[%d]" % i)
        chain.append(dict(fmt=DPPF % lane(K), hs=(yi, ai, zn), writes=yi, dpp_src=ai, reads=(zn, yi), gap=1))
    fill = []
    hs = b.inp("hsym")
    for c in (2 * K, 2 * K + 1):
        q = b.out("Qh[%d]" % c)
        t = b.inp("T[%d]" % c)
        fill.append(dict(fmt="v_add_f64 {0}, {0}, {1}", hs=(q, t), writes=q, dpp_src=None, reads=(q, t)))
    for c in (2 * K, 2 * K + 1):
        q = b.out("Qh[%d]" % c)
        fill.append(dict(fmt="v_mul_f64 {0}, {1}, {0}", hs=(q, hs), writes=q, dpp_src=None, reads=(q, hs)))
    if K + 1 < M:  # the copy the back substitution accumulates into: y[K+1] is final after this column
        pass
    b.lines = interleave(chain, fill)
    sig = ("template <>\nTOLG_DEV void b3_fwd<%d>(double (&y)[6], const double (&a)[6], double nri, double& zn, double (&Qh)[12], "
           "const double (&T)[12], double hsym) {\n" % K)
    return sig + b.render() + "}\n"


def upd_chunk(b, u):
    """Qh[i] += Y[u]@lane i * zn[u] (one rank-1 term of the value update)."""
    ops = []
    for i in range(12):
        acc = b.out("Qh[%d]" % i)
        yu = b.inp("Y[%d]" % u)
        zu = b.inp("zn[%d]" % u)
        ops.append(dict(fmt=DPPF % i, hs=(acc, yu, zu), writes=acc, dpp_src=yu, reads=(zu, acc)))
    return ops


def gen_bwd(K):
    """K = 5..1: nx[K] known -> t[i] += a[K]@lane(i) * nx[K] for i < K (t starts at 0), nx[K-1] = zn[K-1] + nri[K-1] t[K-1];
    rank-1 update 5-K.
    K = 0: the last rank-1 update alone."""
    b = Block()
    if K == 0:
        b.lines = upd_chunk(b, 5)
        sig = ("template <>\nTOLG_DEV void b3_bwd<0>(double (&t)[6], const double (&a)[6], const double (&nri)[6], double (&nx)[6], "
               "double (&Qh)[12], const double (&Y)[6], const double (&zn)[6]) {\n")
        return sig + b.render() + "}\n"
    aK = b.inp("a[%d]" % K)
    nxK = b.inp("nx[%d]" % K)
    chain = []
    order = [K - 1] + [i for i in range(K - 1)]
    tl = {}
    for i in order:
        tl[i] = b.out("t[%d]" % i)
    nri = b.inp("nri[%d]" % (K - 1))
    znp = b.inp("zn[%d]" % (K - 1))
    nxo = b.out("nx[%d]" % (K - 1), "=&v")
    first = True
    for i in order:
        chain.append(dict(fmt=DPPF % lane(i), hs=(tl[i], aK, nxK), writes=tl[i], dpp_src=aK, reads=(nxK, tl[i]), gap=(2 if first else 1)))
        if first:
            chain.append(dict(fmt="v_fma_f64 {0}, {1}, {2}, {3}", hs=(nxo, tl[i], nri, znp), writes=nxo, dpp_src=None,
                              reads=(tl[i], nri, znp), gap=2))
            first = False
    fill = upd_chunk(b, 5 - K)
    b.lines = interleave(chain, fill)
    sig = ("template <>\nTOLG_DEV void b3_bwd<%d>(double (&t)[6], const double (&a)[6], const double (&nri)[6], double (&nx)[6], "
           "double (&Qh)[12], const double (&Y)[6], const double (&zn)[6]) {\n" % K)
    return sig + b.render() + "}\n"


def main():
    out = ["// tolg_k2_blocks.h -- MACHINE-WRITTEN by tools/gen_k2_blocks.py; do not edit.  Interleaved inline-asm blocks of\n"
           "// k_backward3 (m = 6): each statement is one step of a latency chain (factorisation pivot, forward column, backward\n"
           "// column) with its share of the neighbouring product's DPP-fused multiply-adds in the gaps of the chain.\n"
           "// Included by tolg_backward3.h inside namespace tolg.\n",
           "template <bool GRAV, int J>\nTOLG_DEV void b3_piv(double (&a)[6], double& rinv, double& d, double wm, double (&Qh)[12], "
           "const double (&A)[12], const double (&Z)[12]);\n",
           "template <int K>\nTOLG_DEV void b3_fwd(double (&y)[6], const double (&a)[6], double nri, double& zn, double (&Qh)[12], "
           "const double (&T)[12], double hsym);\n",
           "template <int K>\nTOLG_DEV void b3_bwd(double (&t)[6], const double (&a)[6], const double (&nri)[6], double (&nx)[6], "
           "double (&Qh)[12], const double (&Y)[6], const double (&zn)[6]);\n"]
    out.append("#ifndef TOLG_DPP_BUILTIN  // (that diagnostic build runs the plain sequence: K3_PLAIN)\n")
    for grav in (False, True):
        for J in range(M):
            out.append(gen_piv(grav, J))
    for K in range(M):
        out.append(gen_fwd(K))
    for K in range(M - 1, -1, -1):
        out.append(gen_bwd(K))
    out.append("#endif\n")
    with open(OUT, "w") as f:
        f.write("\n".join(out))
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
