import numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads
B, N, K = 4096, 200, 60
prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N)
for mode, ls in (("ss", False), ("ms", True)):
    solver = BatchedTrackingILQR(prob, B)
    r = solver.fit_batch(x0_q, x0_xi, us0, mode=mode, n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0, line_search=ls)
    torch.cuda.synchronize()
    A = r.alpha_hist.cpu().numpy(); it = r.iters.cpu().numpy(); st = r.status.cpu().numpy()
    print(mode, "iters: min %d med %d max %d; status counts" % (it.min(), np.median(it), it.max()), np.bincount(st))
    for k in range(0, K, 3):
        act = it > k
        a = A[act, k]
        last = (it == k + 1) & (st == 2)
        print("it %2d active %4d  alpha==1: %4d  <1: %4d  (of which final/no-descent %4d)  median alpha<1 %.3g" % (
            k, act.sum(), (a == 1.0).sum(), (a < 1.0).sum(), last.sum(), np.median(a[a < 1.0]) if (a < 1.0).any() else 0))
