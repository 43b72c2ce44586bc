import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads
B, N = 512, 200
prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N)
solver = BatchedTrackingILQR(prob, B)
def run(mode, ls, k):
    r = solver.fit_batch(x0_q, x0_xi, us0, mode=mode, n_iterations=k, tol_grad_norm=0.0, tol_d_norm=0.0, line_search=ls)
    torch.cuda.synchronize()
    return r.xs_q.cpu().numpy()[:, :, :3, :3].copy(), r.iters.cpu().numpy().copy()
def ang(Ra, Rb):
    tr = np.einsum("bnij,bnij->bn", Ra, Rb)
    return np.arccos(np.clip((tr - 1) / 2, -1, 1))
ref = prob.q_ref[None, :, :3, :3]
for mode, ls in (("ss", False), ("ms", True)):
    prev, _ = run(mode, ls, 2)
    for k in (3, 5, 8, 12, 18, 24):
        cur, it = run(mode, ls, k)
        if k in (3, 8, 18):
            a = ang(prev_k, cur) if False else None
        prev_k = cur
    # consecutive iterates
    for k in (3, 6, 10, 16, 22):
        Ra, ita = run(mode, ls, k); Rb, itb = run(mode, ls, k + 1)
        live = itb == k + 1
        a = ang(Ra[live], Rb[live])            # deviation nominal -> accepted candidate, per knot
        e = ang(Rb[live], np.broadcast_to(ref, Rb.shape)[live])  # tracking error, per knot
        frac = lambda x, t: float((x.max(axis=0) > t).mean())  # fraction of knots at which SOME trajectory exceeds t (a wave mixes 16)
        print("%s it %2d: live %3d | deviation: knots with any trajectory > 29deg %.2f, > 60deg %.2f, > 90deg %.2f | tracking error: > 29 %.2f, > 60 %.2f, > 90 %.2f"
              % (mode, k, live.sum(), frac(a, 0.506), frac(a, 1.047), frac(a, 1.571), frac(e, 0.506), frac(e, 1.047), frac(e, 1.571)))
