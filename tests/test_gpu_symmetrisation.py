"""The FAST backward sweep symmetrises Q_xx every fourth knot (TOLG_K3_SYMP = 4, csrc/tolg_backward3.h); the reference symmetrises
V at every knot (traopt_controller.py:3004), and so does the full kernel since the end of round 4 (the first sweep of a solve, a single
sweep through linearize_backward, every group the fast sweep hands back: tests/test_gpu_fuzz.py::test_seeds_the_campaigns_flagged is
where the period was outgrown).  tests/test_gpu_k2_fast.py compares the two kernels -- i.e. the two periods -- directly.  The antisymmetric part of V is an unstable mode of the sweep's form of the
recursion, so the period is a numerical choice that needs a bound where it is most exposed: long horizons and small input
weights (the per-knot growth factor is largest there).  Checked against the oracle, which symmetrises every knot: gains,
value-function gradient term and the iterates of a few iterations at N = 400 (drone, R = 1e-5 .. 1e-3) and N = 955 (SE3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import bridge as ob  # noqa: E402  (test infrastructure)
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads  # noqa: E402


def _rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("kind,N,R", [("drone", 400, 1e-3), ("drone", 400, 1e-5), ("se3", 955, 1e-5), ("se3", 400, 1e-6)])
def test_sweep_with_four_knot_symmetrisation_matches_every_knot_oracle(kind, N, R):
    B = 6
    prob, x0_q, x0_xi, us0 = (workloads.drone_tracking if kind == "drone" else workloads.se3_tracking)(B, N=N, R_scale=R)
    op = ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
    # one sweep on the initial multiple-shooting trajectory: gains at every knot of the horizon
    xs_q = np.tile(prob.q_ref[None], (B, 1, 1, 1)); xs_xi = np.tile(prob.xi_ref[None], (B, 1, 1))
    xs_q[:, 0] = x0_q; xs_xi[:, 0] = x0_xi
    r = BatchedTrackingILQR(prob, B).linearize_backward(xs_q, xs_xi, us0, ms=True)
    for b in range(B):
        o = ob.lin_backward(op, xs_q[b], xs_xi[b], us0[b], ms=True)
        Kg, Ko = r["K"][b].cpu().numpy(), o["K"]
        per_knot = np.abs(Kg - Ko).max(axis=(1, 2)) / np.abs(Ko).max(axis=(1, 2))
        assert per_knot.max() < 1e-9, (b, int(per_knot.argmax()), per_knot.max())   # no growth along the horizon
        assert float(r["grad"][b]) == pytest.approx(o["grad"], rel=1e-9)
    # and four accept-always iterations end to end (the sweeps of the second to fourth are the fast kernel's: every fourth knot)
    rr = BatchedTrackingILQR(prob, B).fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=4, tol_grad_norm=0.0, tol_d_norm=0.0)
    oo = ob.fit_batch(op, x0_q, x0_xi, us0, mode="ms", max_iter=4)
    ok = np.isfinite(oo["J_hist"]).all(axis=1) & (np.abs(oo["J_hist"]).max(axis=1) < 1e12)
    assert ok.sum() >= 1
    assert _rel(rr.J_hist.cpu().numpy()[ok], oo["J_hist"][ok]) < 1e-9
