#!/usr/bin/env python3
"""Who is right when the GPU path and the fp64 oracle differ at rounding level?  One seed of tools/parity_fuzz.py, three
evaluations: the GPU (fp64), the oracle (fp64, oracle/tolg_oracle.c) and the oracle's statements in long double
(oracle/tolg_oracle_ld.c, 64-bit mantissa) as the referee.  Prints, per iteration, the largest relative cost distance
GPU-referee and oracle-referee over the batch, the trajectories on which the GPU is much farther from the referee than the
oracle is (a precision loss on the GPU side would show there), every trajectory whose exit code / iteration count differs,
and for those the deciding comparison of the line search as each precision evaluates it.
    python tools/parity_referee.py [--large] [--kinds=3] [--cpu] SEED [trajectory ...]
--cpu: no GPU run (the oracle against its twin only: is this PROBLEM rounding-sensitive?)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_fuzz as pf  # noqa: E402
from oracle import bridge as ob, bridge_ld as obl  # noqa: E402

np.set_printoptions(precision=3, linewidth=220)


def rel(a, c):
    with np.errstate(all="ignore"):
        d = np.abs((np.asarray(a).astype(obl.LD) - c) / c).astype(float)
    return d


def detail(op, cfg, x0_q, x0_xi, us0, b):
    """Single fits of trajectory b in both precisions with the cost of every line-search trial."""
    kw = dict(mode=cfg["mode"], max_iter=cfg["K"], line_search=cfg["line_search"], rollout=cfg["rollout"])
    runs = (("fp64 oracle", ob.fit(op, x0_q[b], x0_xi[b], us0[b], tol_grad=0.0, tol_defect=0.0, **kw)),
            ("long double", obl.fit(op, x0_q[b], x0_xi[b], us0[b], **kw)))
    for name, r in runs:
        print("   %s: iterations %d, status %d, alpha %s, mu %s" % (name, r["n_iters"], r["status"], np.asarray(r["alpha_hist"], float),
                                                                    np.asarray(r["mu_hist"], float)))
        for k in range(min(r["n_iters"] + 1, cfg["K"])):
            n = int(r["n_trials"][k])
            if n:
                print("      it %d  (J_trial - J_opt) / J_opt: %s" % (k, np.asarray((r["trial_J"][k, :n] - r["J_lin"][k]) / r["J_lin"][k], float)))


def main():
    flags = [a for a in sys.argv[1:] if a.startswith("--")]
    args = [int(a) for a in sys.argv[1:] if not a.startswith("--")]
    pf.LARGE = "--large" in flags
    nk = None
    for f in flags:
        if f.startswith("--kinds="):
            nk = int(f.split("=")[1])
    seed, want = args[0], args[1:]
    cfg, prob, x0_q, x0_xi, us0 = pf.draw(seed, nk)
    if isinstance(prob, tuple):
        prob, op = prob
    else:
        op = ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
    K, B = cfg["K"], cfg["B"]
    print("===== seed %d %s  R %s  dt %.4g" % (seed, cfg, np.diag(prob.R), prob.dt))
    kw = dict(mode=cfg["mode"], max_iter=K, line_search=cfg["line_search"], rollout=cfg["rollout"])
    o = ob.fit_batch(op, x0_q, x0_xi, us0, **kw)
    l = obl.fit_batch(op, x0_q, x0_xi, us0, **kw)
    g = None
    if "--cpu" not in flags:
        from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR
        r = BatchedTrackingILQR(prob, B).fit_batch(x0_q, x0_xi, us0, mode=cfg["mode"], n_iterations=K, tol_grad_norm=0.0,
                                                   tol_d_norm=0.0, line_search=cfg["line_search"], rollout=cfg["rollout"])
        g = dict(J_hist=r.J_hist.cpu().numpy(), iters=r.iters.cpu().numpy(), status=r.status.cpu().numpy(), us=r.us.cpu().numpy())
    n = np.minimum(o["iters"], l["iters"])
    if g is not None:
        n = np.minimum(n, g["iters"])
    mask = np.arange(K)[None, :] < n[:, None]
    d_o = np.where(mask, rel(o["J_hist"], l["J_hist"]), 0.0)
    d_o = np.where(np.isfinite(d_o), d_o, 0.0)
    print("oracle fp64 vs long double: max rel J per iteration      %s" % d_o.max(axis=0))
    print("  trajectories with a distance > 1e-12 / 1e-10 / 1e-9: %d / %d / %d of %d"
          % ((d_o.max(axis=1) > 1e-12).sum(), (d_o.max(axis=1) > 1e-10).sum(), (d_o.max(axis=1) > 1e-9).sum(), B))
    fork = np.nonzero((o["iters"] != l["iters"]) | (o["status"] != l["status"]))[0]
    print("  exit code / iteration count differ between the two precisions on %d trajectories %s" % (len(fork), fork[:10]))
    if g is not None:
        d_g = np.where(mask, rel(g["J_hist"], l["J_hist"]), 0.0)
        d_g = np.where(np.isfinite(d_g), d_g, 0.0)
        d_go = np.where(mask, rel(g["J_hist"], o["J_hist"].astype(obl.LD)), 0.0)
        d_go = np.where(np.isfinite(d_go), d_go, 0.0)
        print("GPU vs long double:         max rel J per iteration      %s" % d_g.max(axis=0))
        print("GPU vs oracle fp64:         max rel J per iteration      %s" % d_go.max(axis=0))
        # where is the GPU the outlier?  cumulative distances, GPU more than ten times farther from the referee than the oracle
        cg, co = np.maximum.accumulate(d_g, axis=1), np.maximum.accumulate(d_o, axis=1)
        out = np.nonzero(((cg > 10.0 * co) & (cg > 1e-12)).any(axis=1))[0]
        print("  trajectories on which the GPU is > 10x farther from the referee than the oracle (and > 1e-12): %d %s" % (len(out), out[:10]))
        for b in out[:5]:
            print("    b%d  GPU-ld %s  oracle-ld %s" % (b, d_g[b], d_o[b]))
        worst = np.argsort(-d_go.max(axis=1))[:5]
        print("  the five trajectories with the largest GPU-oracle distance:")
        for b in worst:
            print("    b%d  GPU-oracle %s\n          GPU-ld     %s\n          oracle-ld  %s  status %d/%d/%d" % (b, d_go[b], d_g[b], d_o[b], g["status"][b], o["status"][b], l["status"][b]))
        dis = np.nonzero((g["iters"] != o["iters"]) | (g["status"] != o["status"]))[0]
        print("  exit code / iteration count GPU vs oracle differ on %d trajectories; the referee sides with the GPU on %d, with the oracle on %d"
              % (len(dis), sum(int(l["iters"][b] == g["iters"][b] and l["status"][b] == g["status"][b]) for b in dis),
                 sum(int(l["iters"][b] == o["iters"][b] and l["status"][b] == o["status"][b]) for b in dis)))
        for b in dis[:10]:
            print("    b%d iterations GPU %d / oracle %d / long double %d, status %d / %d / %d" % (b, g["iters"][b], o["iters"][b], l["iters"][b],
                                                                                               g["status"][b], o["status"][b], l["status"][b]))
        want = want or list(dis[:3])
    for b in want:
        print("  -- trajectory %d, line-search trials" % b)
        detail(op, cfg, x0_q, x0_xi, us0, b)


if __name__ == "__main__":
    main()
