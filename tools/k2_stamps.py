"""Per-phase cycle stamps of the backward sweep (debug build: hipcc ... -DTOLG_STAMPS, loaded through
TOLG_HIP_LIB).  Prints s_memtime cycles per knot for each phase of k_backward3 (tolg_backward3.h), wavefront 7."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads

B, N, K = 4096, 200, 12
prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N)
s = BatchedTrackingILQR(prob, B)
r = s.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0)
torch.cuda.synchronize()
st = r.mu_hist[28, :8].cpu().numpy()
names = ["wait for the record DMA, issue ds_reads", "Z = V A (144 dpp fmac, waits for the ds_reads)",
         "lgkm wait, gain stores, record DMA issue", "Qh = L + A^T Z, LDS writes of the symmetrisation",
         "build G / Mt, factorisation (first attempt)", "gradient term, forward substitution",
         "symmetrise, rank-m update, back substitution, gains", "epilogue"]
tot = st[:7].sum()
for n, v in zip(names, st):
    print("%-52s %8.0f cycles/knot  %5.1f %%" % (n, v / N, 100 * v / tot))
print("total per knot %.0f (s_memtime ticks)" % (tot / N))
rt = r.mu_hist[29, :2].cpu().numpy()
print("the sweep of this wave: %.1f us by s_memrealtime (100 MHz), %.0f s_memtime ticks -> %.2f GHz" % (rt[0] / 100.0, rt[1], rt[1] / (rt[0] / 100.0) / 1e3))
