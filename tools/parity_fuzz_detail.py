"""Cost / step-size / gradient / defect histories GPU vs oracle for seeds of tools/parity_fuzz.py.
    python tools/parity_fuzz_detail.py SEED..."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),"tools"))
import numpy as np
import parity_fuzz as pf
from oracle import bridge as ob
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR
np.set_printoptions(precision=6, linewidth=200)
for seed in [int(a) for a in sys.argv[1:]]:
    cfg, prob, x0_q, x0_xi, us0 = pf.draw(seed)
    K, B = cfg["K"], cfg["B"]
    print("=====", seed, cfg, "dt", (prob[0] if isinstance(prob, tuple) else prob).dt)
    solver = BatchedTrackingILQR(prob[0] if isinstance(prob, tuple) else prob, B)
    r = solver.fit_batch(x0_q, x0_xi, us0, mode=cfg["mode"], n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0, line_search=cfg["line_search"], rollout=cfg["rollout"])
    op = prob[1] if isinstance(prob, tuple) else ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
    prob = prob[0] if isinstance(prob, tuple) else prob
    o = ob.fit_batch(op, x0_q, x0_xi, us0, mode=cfg["mode"], max_iter=K, line_search=cfg["line_search"], rollout=cfg["rollout"])
    it, st, Jg = r.iters.cpu().numpy(), r.status.cpu().numpy(), r.J_hist.cpu().numpy()
    Ag = r.alpha_hist.cpu().numpy() if r.alpha_hist is not None else None
    for b in range(min(B,3)):
        print(" b", b, "iters", it[b], o["iters"][b], "status", st[b], o["status"][b])
        print("   Jg", Jg[b,:max(it[b],1)+1])
        print("   Jo", o["J_hist"][b,:max(o["iters"][b],1)+1])
        if Ag is not None: print("   Ag", Ag[b,:it[b]+1], "\n   Ao", o.get("alpha_hist", np.zeros((B,1)))[b,:o["iters"][b]+1] if "alpha_hist" in o else None)
        for k in ("grad_hist","defect_hist","mu_hist"):
            g = getattr(r,k,None)
            if g is not None and k in o: print("  ",k,"g",g[b,:it[b]+1].cpu().numpy(),"\n  ",k,"o",o[k][b,:o["iters"][b]+1])
