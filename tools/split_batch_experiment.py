"""Experiment: the 4096 batch as two independent 2048 halves on two HIP streams (kernels of one half overlap
kernels of the other) against the single-stream solve.  PYTHONPATH=. python tools/split_batch_experiment.py"""
import time
import torch
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads

B, N, W, K = 4096, 200, 3, 20
prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N)
dev = torch.device("cuda:0")


def run(parts):
    n = B // parts
    solvers, streams = [], []
    for p in range(parts):
        s = BatchedTrackingILQR(prob, n, device=dev)
        st = torch.cuda.Stream(device=dev)
        sl = slice(p * n, (p + 1) * n)
        with torch.cuda.stream(st):
            s.solve_begin(torch.as_tensor(x0_q[sl], device=dev), torch.as_tensor(x0_xi[sl], device=dev),
                          torch.as_tensor(us0[sl], device=dev), mode="ms", n_iterations=W + K, tol_grad_norm=0.0, tol_d_norm=0.0)
            s.solve_iterate(W)
        solvers.append(s); streams.append(st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(K):  # interleave the launches so that neither stream runs ahead by a whole solve
        for s, st in zip(solvers, streams):
            with torch.cuda.stream(st):
                s.solve_iterate(1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for s, st in zip(solvers, streams):
        with torch.cuda.stream(st):
            s.solve_end()
    torch.cuda.synchronize()
    return K / dt


for parts in (1, 2, 4):
    print("parts", parts, "batch-iterations/s %.1f" % run(parts))
