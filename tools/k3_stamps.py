"""Per-phase cycle stamps of the rollout (debug build: hipcc ... -DTOLG_STAMPS, loaded through TOLG_HIP_LIB)."""
import sys

import torch
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads

B, N, K = 4096, 200, 12
prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N)
s = BatchedTrackingILQR(prob, B)
sched = sys.argv[1] if len(sys.argv) > 1 else "auto"   # auto: fused rollout + linearisation; split: rollout alone
r = s.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0, schedule=sched)
torch.cuda.synchronize()
st = r.alpha_hist[80, :8].cpu().numpy()
names = ["loop overhead + next-state prefetch issue", "issue gains / controls / factors loads", "Log(x^-1 x_new)", "K dx + quad broadcast",
         "dynamics (Exp, inertia)", "compose + project", "stores (split) / hand-over", "publish to the LDS ring (fused)"]
tot = st[:8].sum()
print("schedule:", sched)
for n, v in zip(names, st):
    print("%-45s %8.0f cycles/knot  %5.1f %%" % (n, v / N, 100 * v / tot))
print("total per knot %.0f (s_memtime ticks)" % (tot / N))
if sched == "auto":
    for h in range(2):
        w, k, n = r.alpha_hist[81 + h, :3].cpu().numpy()
        print("helper %d: %d passes, %.0f cycles per pass working, %.0f waiting for the rollout" % (h, n, k / max(n, 1), w / max(n, 1)))
