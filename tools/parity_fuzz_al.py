#!/usr/bin/env python3
"""Randomised parity sweep of the augmented-Lagrangian outer loop (AL_iLQR_Tracking_SE3_MS, traopt_controller.py:3218-3293)
on the GPU against the outer loop restated around the oracle (tests/test_gpu_parity.py::_al_oracle -- the reference class
does not run at HEAD, SURVEY App. C-Q7, so parity is unpinned here as everywhere for this class): random constant-twist
references, weights, box bounds that the unconstrained solution violates, penalty schedules, horizons, batches.
    python tools/parity_fuzz_al.py [cases] [first seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_parity import _al_oracle  # noqa: E402
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, TrackingProblem, workloads  # noqa: E402
from trajectory_optimization_matrix_lie_groups_amd.workloads import _se3_exp  # noqa: E402


def one(seed):
    rng = np.random.default_rng(seed)
    N, B = int(rng.integers(10, 60)), int(rng.integers(1, 6))
    dt = float(10.0 ** rng.uniform(-2.2, -1.5))
    xi_c = np.r_[rng.normal(size=3) * 0.7, rng.normal(size=3) * 1.5]
    step = _se3_exp(xi_c * dt)
    q_ref = np.empty((N + 1, 4, 4)); q_ref[0] = np.eye(4)
    for i in range(N):
        q_ref[i + 1] = q_ref[i] @ step
    xi_ref = np.repeat(xi_c[None], N + 1, 0)
    Q = np.diag(10.0 ** rng.uniform(-0.5, 1.5, 12))
    R = np.eye(6) * 10.0 ** rng.uniform(-4, -2)
    prob = TrackingProblem("se3", np.diag(rng.uniform(0.4, 1.5, 6)), dt, Q, R, rng.uniform(1, 10) * Q, q_ref, xi_ref)
    q0 = np.eye(4); q0[:3, 3] = rng.normal(size=3) * 0.3
    xi0 = xi_c + rng.normal(size=6) * 0.2
    x0_q, x0_xi = workloads.perturbed_batch(q0, xi0, B, 0.1 * np.ones(6), 0.05, seed=seed)
    us0 = np.zeros((B, N, 6))
    solver = BatchedTrackingILQR(prob, B)
    n_in = int(rng.integers(20, 60))
    free = solver.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=n_in)
    umax = float(free.us.abs().max())
    if not np.isfinite(umax) or umax < 1e-3:
        return None
    bound = umax * rng.uniform(0.3, 0.8)            # the unconstrained solution violates the box
    lb, ub = -bound * np.ones(6), bound * np.ones(6) * rng.uniform(0.8, 1.2)
    n_al = int(rng.integers(2, 7))
    kw = dict(tol_constr=10.0 ** rng.uniform(-3, -1) * bound, mu0=10.0 ** rng.uniform(-3, -1), mu_scale=float(rng.uniform(3, 20)))
    res, info = solver.al_fit_batch(x0_q, x0_xi, us0, lb, ub, n_al_iters=n_al, n_ilqr_iters=n_in, **kw)
    worst = dict(J=0.0, u=0.0, lam=0.0)
    notes = []
    for b in range(B):
        o, lam, imu, mu, n_outer = _al_oracle(prob, x0_q[b], x0_xi[b], us0[b], lb, ub, n_al, n_in, kw["tol_constr"], mu0=kw["mu0"],
                                              mu_scale=kw["mu_scale"])
        n = int(res.iters[b])
        if n != o["n_iters"]:
            notes.append("b%d inner iterations of the last solve %d/%d" % (b, n, o["n_iters"]))
            continue
        Jg = res.J_hist[b, :n].cpu().numpy()
        worst["J"] = max(worst["J"], np.abs(Jg - o["J_hist"][:n]).max() / np.abs(o["J_hist"][:n]).max())
        worst["u"] = max(worst["u"], np.abs(res.us[b].cpu().numpy() - o["us"]).max() / max(1.0, np.abs(o["us"]).max()))
        worst["lam"] = max(worst["lam"], np.abs(info["lmbd"][b].cpu().numpy() - lam).max() / max(1.0, np.abs(lam).max()))
        if abs(float(info["mu"][b]) - mu) > 1e-12 * mu:
            notes.append("b%d mu %.3e/%.3e" % (b, float(info["mu"][b]), mu))
    return dict(N=N, B=B, n_al=n_al, n_in=n_in, bound=round(bound, 3), outer=int(info["outer_iterations"])), worst, notes


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    bad = skipped = 0
    for seed in range(s0, s0 + n):
        r = one(seed)
        if r is None:
            skipped += 1
            continue
        cfg, w, notes = r
        flag = w["J"] > 1e-8 or w["u"] > 1e-6 or w["lam"] > 1e-6 or notes
        bad += bool(flag)
        print("%s seed %d %s  J %.1e u %.1e lambda %.1e  %s" % ("DIFF" if flag else "ok  ", seed, cfg, w["J"], w["u"], w["lam"], "; ".join(notes)), flush=True)
    print("%d of %d cases differ (%d skipped)" % (bad, n - skipped, skipped))


if __name__ == "__main__":
    main()
