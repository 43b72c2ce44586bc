"""Per-iteration relative differences GPU vs oracle for one seed of tools/parity_fuzz.py, both schedules.
    python tools/parity_fuzz_reldiff.py SEED"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),"tools"))
import numpy as np
import parity_fuzz as pf
from oracle import bridge as ob
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR
np.set_printoptions(precision=3, linewidth=220)
seed=int(sys.argv[1])
cfg, prob, x0_q, x0_xi, us0 = pf.draw(seed)
K,B=cfg["K"],cfg["B"]
op0 = prob[1] if isinstance(prob, tuple) else None   # SO(3) / pendulum cases bring their oracle problem along
prob = prob[0] if isinstance(prob, tuple) else prob
for sched in ("auto","split"):
    solver = BatchedTrackingILQR(prob[0] if isinstance(prob, tuple) else prob, B)
    r = solver.fit_batch(x0_q, x0_xi, us0, mode=cfg["mode"], n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0, line_search=cfg["line_search"], rollout=cfg["rollout"], schedule=sched)
    op = op0 if op0 is not None else ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
    o = ob.fit_batch(op, x0_q, x0_xi, us0, mode=cfg["mode"], max_iter=K, line_search=cfg["line_search"], rollout=cfg["rollout"])
    Jg=r.J_hist.cpu().numpy(); it=r.iters.cpu().numpy(); A=r.alpha_hist.cpu().numpy(); dg=r.defect_hist.cpu().numpy(); gg=r.grad_hist.cpu().numpy()
    print("schedule", sched)
    for b in range(B):
        n=min(it[b],o["iters"][b])
        print(b, it[b], o["iters"][b], "relJ", np.abs(Jg[b,:n]-o["J_hist"][b,:n])/np.abs(o["J_hist"][b,:n]), "alpha", A[b,:n])
        print("     rel defect", np.abs(dg[b,:n+1]-o["defect_hist"][b,:n+1])/np.abs(o["defect_hist"][b,:n+1]), "rel grad", np.abs(gg[b,:n]-o["grad_hist"][b,:n])/np.abs(o["grad_hist"][b,:n]))
