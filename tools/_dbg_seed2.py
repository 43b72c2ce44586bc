import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_fuzz as pf
from oracle import bridge as ob
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR
np.set_printoptions(precision=4, linewidth=220)
seed, K8, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg, prob, x0_q, x0_xi, us0 = pf.draw(seed, None)
if isinstance(prob, tuple):
    prob, op = prob
else:
    op = ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
B = cfg["B"]
s = BatchedTrackingILQR(prob, B)
r = s.fit_batch(x0_q, x0_xi, us0, mode=cfg["mode"], n_iterations=K8, tol_grad_norm=0.0, tol_d_norm=0.0, line_search=cfg["line_search"], rollout=cfg["rollout"])
xs_q, xs_xi, us = r.xs_q.cpu().numpy(), r.xs_xi.cpu().numpy(), r.us.cpu().numpy()
print("state magnitudes: |xi| max", np.abs(xs_xi[b]).max(), " |us| max", np.abs(us[b]).max(), " pos max", np.abs(xs_q[b][:, :3, 3]).max())
g = s.linearize_backward(xs_q, xs_xi, us, ms=True)
o = ob.lin_backward(op, xs_q[b], xs_xi[b], us[b], ms=True)
for k in ("Fx", "d", "lx", "K", "k"):
    a = g[k][b].cpu().numpy(); c = np.asarray(o[{"lx": "Lx"}.get(k, k)])
    print(k, "GPU finite", np.isfinite(a).all(), "oracle finite", np.isfinite(c).all(), "max|GPU|", np.nanmax(np.abs(a)), "max|oracle|", np.nanmax(np.abs(c)),
          "rel diff", np.nanmax(np.abs(a - c)) / max(np.nanmax(np.abs(c)), 1e-300))
print("J", float(g["J"][b]), o["J"], "grad", float(g["grad"][b]), o["grad"], "mu GPU", float(g["mu"][b]) if "mu" in g else None, "oracle mu", o.get("mu"))
print(list(g.keys()))
