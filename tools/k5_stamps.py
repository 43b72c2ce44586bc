"""Per-phase cycle stamps of wave B of the wave-pair backward sweep (debug build: hipcc ... -DTOLG_STAMPS5, loaded through
TOLG_HIP_LIB).  s_memtime cycles per knot for each phase of k_backward5 (tolg_backward5.h), workgroup 7."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads  # noqa: E402

B, N, K = 4096, 200, 12
prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N)
s = BatchedTrackingILQR(prob, B)
r = s.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0)
torch.cuda.synchronize()
st = r.mu_hist[28, :8].cpu().numpy()
names = ["LDS reads of the records and constants issued", "Z = V [F_x | d], rows 6.. (72 dpp fmac; waits for the reads)",
         "barrier 1 (lgkmcnt(0) + s_barrier)", "gain stores, exchange read, Q_xx rows 6.. (63 dpp fmac)",
         "G, Mt, factorisation, PD vote", "gradient term, forward substitution, Y | zn to LDS",
         "barrier 2", "update, symmetrisation, back substitution, gains"]
tot = st.sum()
for n, v in zip(names, st):
    print("%-62s %8.0f cycles/knot  %5.1f %%" % (n, v / N, 100 * v / tot))
print("total per knot %.0f (s_memtime ticks)" % (tot / N))
rt = r.mu_hist[29, :2].cpu().numpy()
print("loop of wave B: %.1f us by s_memrealtime (100 MHz), %.0f s_memtime ticks -> %.2f GHz" % (rt[0] / 100.0, rt[1], rt[1] / (rt[0] / 100.0) / 1e3))
