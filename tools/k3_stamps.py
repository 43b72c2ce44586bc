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
if sched == "auto":  # roll_step_twist in wave 0 of k_rollout_lin
    names = ["loop overhead", "issue gains / controls reads (LDS)", "compose + Log(x^-1 x_new)", "K dx + quad broadcast",
             "twist half of the dynamics", "pose hand-over from the pose wave", "(unused)",
             "look-ahead (back-pressure, inputs, next nominal state) + publish + pose read"]
else:               # roll_step in k_rollout
    names = ["loop overhead + next-state prefetch issue", "issue gains / controls / factors loads", "Log(x^-1 x_new) + pose half",
             "K dx + quad broadcast", "twist half of the dynamics", "compose + project", "stores", "(unused)"]
tot = st[:8].sum()
print("schedule:", sched)
for n, v in zip(names, st):
    print("%-45s %8.0f cycles/knot  %5.1f %%" % (n, v / N, 100 * v / tot))
print("total per knot %.0f (s_memtime ticks)" % (tot / N))
if sched == "auto":
    pw = r.alpha_hist[83, :3].cpu().numpy() / N
    print("pose wave per knot: %.0f waiting for the twist, %.0f input DMA issue + publish, %.0f pose chain + publish" % tuple(pw))
    for h in range(2):
        w, k, n = r.alpha_hist[81 + h, :3].cpu().numpy()
        print("helper %d: %d passes, %.0f cycles per pass working, %.0f waiting for the rollout" % (h, n, k / max(n, 1), w / max(n, 1)))
