#!/usr/bin/env python3
"""Is a batch beyond one wave per SIMD (4096 trajectories) better served by ONE launch per kernel over all of it, or by
back-to-back launches over 4096-trajectory chunks (separate solver handles on the same stream)?
    python tools/chunked_batch_experiment.py [B] [chunk]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K, W, R = 10, 3, 5
prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=200)
dev = torch.device("cuda")
kw = dict(mode="ms", tol_grad_norm=0.0, tol_d_norm=0.0, n_iterations=W + R * K)


def run(solvers, parts):
    for s, (lo, hi) in zip(solvers, parts):
        s.solve_begin(torch.as_tensor(x0_q[lo:hi], device=dev), torch.as_tensor(x0_xi[lo:hi], device=dev),
                      torch.as_tensor(us0[lo:hi], device=dev), **kw)
    for _ in range(W):
        for s in solvers:
            s.solve_iterate(1)
    ts = []
    for _ in range(R):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            for s in solvers:
                s.solve_iterate(1)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e3)
    res = [s.solve_end() for s in solvers]
    ok = all(bool(torch.isfinite(r.J_hist).all()) for r in res)
    return sorted(ts)[len(ts) // 2], ok


one = [BatchedTrackingILQR(prob, B)]
t1, ok1 = run(one, [(0, B)])
del one
parts = [(lo, min(lo + C, B)) for lo in range(0, B, C)]
many = [BatchedTrackingILQR(prob, hi - lo) for lo, hi in parts]
t2, ok2 = run(many, parts)
print("B %d: one launch per kernel %.3f ms per iteration (%.2f M trajectory-iterations/s, finite %s); %d chunks of %d back to back "
      "%.3f ms (%.2f M, finite %s)" % (B, t1, B / t1 / 1e3, ok1, len(parts), C, t2, B / t2 / 1e3, ok2))
