#!/usr/bin/env python3
"""Soak of the line-search modes: many solves back to back at the benchmark size, every status checked for TOLG_ST_INTERNAL (4: a
wavefront of a ring-synchronised kernel -- the fused launch, k_rollout_ls2 -- gave up waiting for its partner; never expected) and
the results of repeated identical solves compared bit for bit (the kernels are deterministic).
    python tools/soak_ls.py [solves per mode] [iterations]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads  # noqa: E402

n_solves = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B, N = 4096, 200
prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N)
for name, kw in (("merit", dict(mode="ms", line_search=True)), ("ss", dict(mode="ss")), ("linear merit", dict(mode="ms", line_search=True, rollout="linear")),
                 ("accept-always", dict(mode="ms"))):
    solver = BatchedTrackingILQR(prob, B)
    ref = None
    t0 = time.time()
    internal = 0
    same = True
    for s in range(n_solves):
        r = solver.fit_batch(x0_q, x0_xi, us0, n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0, **kw)
        st = r.status.cpu().numpy()
        internal += int((st == 4).sum())
        J = r.J_hist.cpu().numpy()
        if ref is None:
            ref = J.copy()
        else:
            same = same and np.array_equal(np.nan_to_num(J), np.nan_to_num(ref))
    torch.cuda.synchronize()
    print("%-14s %d solves x %d iterations of %d x %d in %.1f s: TOLG_ST_INTERNAL on %d trajectories; repeated solves bitwise equal: %s; statuses of the last: %s"
          % (name, n_solves, K, B, N, time.time() - t0, internal, same, np.bincount(st, minlength=5)), flush=True)
