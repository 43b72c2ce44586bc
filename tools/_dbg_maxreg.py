"""Unit-level look at a max-regularised backward sweep (round 4, seeds 2005 / 20260 of tools/parity_fuzz.py): the first sweep of the
seed's problem through tolg_linearize_backward against the oracle's lin_backward -- regularisation reached, gradient term, and
the knots at which the gains differ by more than 1e-6.    python tools/_dbg_maxreg.py SEED"""
import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
np.set_printoptions(precision=3, linewidth=220)
import parity_fuzz as pf
from oracle import bridge as ob
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2005
cfg, prob, x0_q, x0_xi, us0 = pf.draw(seed)
B, N = cfg["B"], cfg["N"]
op = ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
ms = cfg["mode"] == "ms"
# the trajectory the first backward pass sees: SS = open-loop rollout of us0, MS = reference path with x0 in front
xs_q = np.zeros((B, N + 1, 4, 4)); xs_xi = np.zeros((B, N + 1, 6))
for b in range(B):
    if ms:
        xs_q[b] = prob.q_ref; xs_xi[b] = prob.xi_ref; xs_q[b, 0] = x0_q[b]; xs_xi[b, 0] = x0_xi[b]
    else:
        q, xi = x0_q[b], x0_xi[b]
        xs_q[b, 0], xs_xi[b, 0] = q, xi
        for i in range(N):
            q, xi = ob.f(op, q, xi, us0[b, i]); xs_q[b, i + 1], xs_xi[b, i + 1] = q, xi
print(cfg, "max |xi| per trajectory", np.abs(xs_xi).max(axis=(1, 2)))
solver = BatchedTrackingILQR(prob, B)
r = solver.linearize_backward(xs_q, xs_xi, us0, ms=ms)
for b in range(B):
    o = ob.lin_backward(op, xs_q[b], xs_xi[b], us0[b], ms=ms)
    Kg, Ko = r["K"][b].cpu().numpy(), o["K"]
    kg, ko = r["k"][b].cpu().numpy(), o["k"]
    relK = np.abs(Kg - Ko).max(axis=(1, 2)) / np.maximum(np.abs(Ko).max(axis=(1, 2)), 1e-300)
    relk = np.abs(kg - ko).max(axis=1) / np.maximum(np.abs(ko).max(axis=1), 1e-300)
    bad = np.nonzero(relK > 1e-6)[0]
    print("b%d mu GPU %.3e oracle %.3e delta %.3e/%.3e grad %.3e/%.3e  first knot (from the end) with rel K error > 1e-6: %s  max|V| proxy |K|max %.2e"
          % (b, float(r["mu_delta"][b, 0]), o["mu"], float(r["mu_delta"][b, 1]), o["delta"], float(r["grad"][b]), o["grad"],
             (int(bad.max()) if bad.size else None), np.abs(Ko).max()))
    if bad.size:
        i = int(bad.max())
        print("    rel K error knots %d..%d: %s" % (max(i - 3, 0), min(i + 2, N - 1), relK[max(i - 3, 0): i + 3]), " rel k:", relk[max(i - 3, 0): i + 3])
        print("    |Vxx| oracle at knots around:", [float(np.abs(o["Lxx"][j]).max()) for j in range(max(i - 1, 0), min(i + 3, N + 1))])
