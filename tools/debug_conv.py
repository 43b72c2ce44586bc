import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import bridge as ob
from trajectory_optimization_matrix_lie_groups_amd import workloads, BatchedTrackingILQR
np.set_printoptions(linewidth=250, precision=3)
B=8
prob,x0_q,x0_xi,us0 = workloads.se3_tracking(B,N=60,R_scale=1e-3)
op = ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
o = ob.fit_batch(op,x0_q,x0_xi,us0,mode='ms',max_iter=60,tol_grad=1e-7,tol_defect=1e-6)
s = BatchedTrackingILQR(prob,B)
r = s.fit_batch(x0_q,x0_xi,us0,mode='ms',n_iterations=60,tol_grad_norm=1e-7)
G=r.grad_hist.cpu().numpy(); D=r.defect_hist.cpu().numpy(); J=r.J_hist.cpu().numpy()
for b in [0,3,5]:
    print('b',b,'iters',int(r.iters[b]),o['iters'][b])
    print(' gpu grad', G[b][20:45]); print(' cpu grad', o['grad_hist'][b][20:45])
    print(' gpu def ', D[b][20:45]); print(' cpu def ', o['defect_hist'][b][20:45])
# single-trajectory run of b=3 alone
r1 = BatchedTrackingILQR(prob,1).fit_batch(x0_q[3:4],x0_xi[3:4],us0[3:4],mode='ms',n_iterations=60,tol_grad_norm=1e-7)
print('b3 alone iters', int(r1.iters[0]), r1.grad_hist.cpu().numpy()[0][20:45])
