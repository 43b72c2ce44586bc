#!/usr/bin/env python3
"""Which series tier do the knots of a solve run in, iteration window by iteration window?  (is the
headline's first timed region slower because of clocks or because of WORK?)  Needs a -DTOLG_TIER_COUNT build:
    python tools/build_ab.py tier -DTOLG_TIER_COUNT
    TOLG_HIP_LIB=build_ab/libtolg_tier.so python tools/tier_share.py [--mode ms|ss] [--line-search] [--batch 4096] [--horizon 200]
Every Exp / Log / Jacobian-coefficient evaluation of tolg_lie.h runs under a SeriesGate: short series (per-step rotation
< 0.2 rad, deviation from the nominal < 0.06 rad), long series, or the closed forms for the lanes outside the series' domain;
a wave pays for the most expensive tier any of its lanes needs.  The counters are per lane and per gate (wave x step); the
table prints both shares per window of iterations of ONE fresh solve.  (No times: every gate of the diagnostic build adds to
six global counters with atomics, which serialises the chip -- 5.9 ms per iteration instead of 0.57; the time per window of the
product build is bench.py's `fresh_solve` series.)"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, _capi, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="ms")
    ap.add_argument("--line-search", action="store_true")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=200)
    ap.add_argument("--windows", default="0,1,2,3,5,10,15,25,45,65,105,205")
    a = ap.parse_args()
    lib = _capi.load()
    if not hasattr(lib, "tolg_debug_tier_counts"):
        raise SystemExit("this library has no tier counters: build with -DTOLG_TIER_COUNT and point TOLG_HIP_LIB at it")
    lib.tolg_debug_tier_counts.restype = C.c_int
    lib.tolg_debug_tier_counts.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    edges = [int(x) for x in a.windows.split(",")]
    prob, x0_q, x0_xi, us0 = workloads.se3_tracking(a.batch, N=a.horizon)
    dev = torch.device("cuda", 0)
    solver = BatchedTrackingILQR(prob, a.batch, device=dev)
    x0_q_d, x0_xi_d, us0_d = (torch.as_tensor(v, device=dev) for v in (x0_q, x0_xi, us0))
    for rep in range(2):   # the first pass warms the GPU up; the second is printed
        solver.solve_begin(x0_q_d, x0_xi_d, us0_d, mode=a.mode, n_iterations=edges[-1], tol_grad_norm=0.0, tol_d_norm=0.0,
                           line_search=a.line_search)
        out = (C.c_ulonglong * 6)()
        lib.tolg_debug_tier_counts(out, 1)
        rows = []
        for lo, hi in zip(edges[:-1], edges[1:]):
            solver.solve_iterate(hi - lo)
            torch.cuda.synchronize(dev)
            dt = 0.0
            lib.tolg_debug_tier_counts(out, 1)
            c = np.array(list(out), dtype=float)
            rows.append((lo, hi, dt, c[:3] / max(c[:3].sum(), 1), c[3:] / max(c[3:].sum(), 1)))
        res = solver.solve_end()
    print("# %s%s, %d x %d, one fresh solve (second pass, GPU warm); shares of lanes / of gates (wave x step) per tier"
          % (a.mode, " + line search" if a.line_search else "", a.batch, a.horizon))
    print("# iterations     lanes: short  long  closed-form   gates: short  long  closed-form")
    for lo, hi, dt, l, g in rows:
        print("  %4d..%-4d       %.4f %.4f %.4f            %.4f %.4f %.4f" % (lo, hi - 1, l[0], l[1], l[2], g[0], g[1], g[2]))
    print("# active at the end: %.4f" % float((res.iters == edges[-1]).double().mean().item()))


if __name__ == "__main__":
    main()
