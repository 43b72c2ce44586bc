#!/usr/bin/env python3
"""How many members of the drone N = 400 batch (BASELINE config 5) does accept-always MS keep (status 0, finite) over a
long solve, as a function of the spread of the perturbed initial states?  Decides the spread bench.py --workload drone400
uses: a throughput figure is only valid while every trajectory does full work.
    python tools/drone_spread_survival.py [B] [iterations] [horizon]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 210
NH = int(sys.argv[3]) if len(sys.argv) > 3 else 400
for R_scale, spread in [(r, s) for r in (1e-5, 1e-4, 1e-3) for s in (1.0, 0.3, 0.1)]:
    prob, x0_q, x0_xi, us0 = workloads.drone_tracking(B, N=NH, perturb=spread, R_scale=R_scale)
    solver = BatchedTrackingILQR(prob, B)
    r = solver.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0)
    st = r.status.cpu().numpy()
    it = r.iters.cpu().numpy()
    bad = np.where(st != 0)[0]
    J = r.J_hist.cpu().numpy()
    ok = st == 0
    print("N %d R %.0e spread %.2f: ok %d / %d, statuses %s, first failure at iteration %s, median final J of the ok ones %.4g, J[0] median %.4g"
          % (NH, R_scale, spread, ok.sum(), B, dict(zip(*np.unique(st, return_counts=True))), it[bad].min() if bad.size else None,
             np.median(J[ok, K - 1]) if ok.any() else float("nan"), np.median(J[:, 0])), flush=True)
    # the nominal member and where the failures start
    if bad.size:
        b = bad[np.argmin(it[bad])]
        print("   member %d fails after %d iterations: J %s" % (b, it[b], np.array2string(J[b, max(0, it[b] - 4): it[b] + 1], precision=4)), flush=True)
    del solver
