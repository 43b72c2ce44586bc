"""BASELINE.json configs 2-5 at their stated sizes on one MI355X, each checked against the oracle on a
sample of batch members to the north_star tolerance (<= 1e-6 relative).  The metric configuration
(4096 x 200) is covered by bench.py and tests/test_gpu_parity.py::test_full_size_properties_4096x200."""
import numpy as np
import pytest

from oracle import bridge as ob
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads
from tests.test_gpu_parity import _al_oracle, _oracle_problem, _rel

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _check_members(r, prob, x0_q, x0_xi, us0, members, K, mode="ms", **kw):
    op = _oracle_problem(prob)
    for b in members:
        o = ob.fit(op, x0_q[b], x0_xi[b], us0[b], mode=mode, max_iter=K, **kw)
        n = int(r.iters[b])
        assert n == o["n_iters"]
        assert _rel(r.J_hist[b, :n].cpu(), o["J_hist"][:n]) < TOL
        assert _rel(r.us[b].cpu(), o["us"]) < TOL
        assert _rel(r.xs_xi[b].cpu(), o["xs_xi"]) < TOL
        assert np.abs(r.xs_q[b].cpu().numpy() - o["xs_q"]).max() < TOL


def test_config2_so3_exact_tracking_b1_n100():
    """main_SO3ddp_tracking_exact.py: SS solver, 200 iterations, default tol_grad_norm = 1e-6."""
    prob, x0_q, x0_xi, us0 = workloads.so3_tracking(4, N=100)
    solver = BatchedTrackingILQR(prob, 4)
    r = solver.fit_batch(x0_q, x0_xi, us0, mode="ss", n_iterations=200, tol_grad_norm=1e-6)
    op = ob.embed_so3_problem(prob.J[:3, :3], prob.dt, np.diag([10.0, 10, 10, 1, 1, 1]), prob.R[:3, :3],
                              np.diag([100.0, 100, 100, 10, 10, 10]), prob.q_ref[:, :3, :3], prob.xi_ref[:, :3])
    for b in range(4):  # member 0 is the reference's own B = 1 problem
        o = ob.fit(op, x0_q[b], x0_xi[b], us0[b], mode="ss", max_iter=200, tol_grad=1e-6)
        n = int(r.iters[b])
        assert n == o["n_iters"] and bool(r.converged[b]) == o["converged"]
        assert _rel(r.J_hist[b, :n].cpu(), o["J_hist"][:n]) < TOL
        assert _rel(r.us[b].cpu(), o["us"]) < TOL


def test_config3_se3_exact_tracking_b256_n200():
    B, K = 256, 12
    prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=200)
    solver = BatchedTrackingILQR(prob, B)
    r = solver.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0)
    assert bool(torch_isfinite(r))
    _check_members(r, prob, x0_q, x0_xi, us0, [0, 1, 77, 130, 255], K, tol_grad=0.0, tol_defect=0.0)


def test_config4_al_ddp_input_box_b1024_n200():
    B = 1024
    prob, x0_q, x0_xi, us0, lb, ub = workloads.al_tracking(B, N=200)
    solver = BatchedTrackingILQR(prob, B)
    free = solver.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=30)
    assert float(free.us.abs().max()) > 10.5  # the box is active, otherwise the config is vacuous
    n_al, n_in = 6, 30
    res, info = solver.al_fit_batch(x0_q, x0_xi, us0, lb, ub, n_al_iters=n_al, n_ilqr_iters=n_in, tol_constr=1e-2)
    # the multipliers pull the inputs towards the box (how far after six outer iterations is the algorithm's
    # business, not a parity statement: the parity statement is the per-member comparison below)
    viol_free = (free.us.abs() - 10.0).clamp(min=0).amax(dim=(1, 2))
    viol_al = (res.us.abs() - 10.0).clamp(min=0).amax(dim=(1, 2))
    assert float(viol_al.median()) < 0.7 * float(viol_free.median())
    for b in (0, 3, 500, 1023):
        o, lam, imu, mu, n_outer = _al_oracle(prob, x0_q[b], x0_xi[b], us0[b], lb, ub, n_al, n_in, 1e-2)
        assert _rel(res.us[b].cpu(), o["us"]) < TOL
        assert _rel(res.xs_xi[b].cpu(), o["xs_xi"]) < TOL
        assert _rel(info["lmbd"][b].cpu(), lam) < TOL
        assert float(info["mu"][b]) == pytest.approx(mu)


def test_config5_drone_racing_shard_b1024_n400():
    """One rank's share of the 8192-trajectory batch (8 x 1024)."""
    from trajectory_optimization_matrix_lie_groups_amd.sharding import shard_bounds
    Btot, world, K = 8192, 8, 10
    lo, hi = shard_bounds(Btot, world, 3)
    assert hi - lo == 1024
    prob, x0_q, x0_xi, us0 = workloads.drone_tracking(Btot, N=400)
    x0_q, x0_xi, us0 = x0_q[lo:hi], x0_xi[lo:hi], us0[lo:hi]
    solver = BatchedTrackingILQR(prob, hi - lo)
    r = solver.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0)
    # accept-always MS (line_search=False, benchmark_drone_racing_tracking.py:217) blows up on some of the
    # perturbed members; the reference algorithm does too -- the same members, with the same status
    o = ob.fit_batch(_oracle_problem(prob), x0_q[:256], x0_xi[:256], us0[:256], mode="ms", max_iter=K)
    st = r.status.cpu().numpy()
    np.testing.assert_array_equal(st[:256], o["status"])
    np.testing.assert_array_equal(r.iters.cpu().numpy()[:256], o["iters"])
    ok = np.where(o["status"] == 0)[0]
    assert 200 < ok.size < 256
    import torch
    good = torch.as_tensor(st == 0, device=r.us.device)
    assert torch.isfinite(r.us[good]).all() and torch.isfinite(r.xs_xi[good]).all()
    for b in (ok[0], ok[len(ok) // 2], ok[-1]):
        assert _rel(r.J_hist[b].cpu(), o["J_hist"][b]) < TOL
        assert _rel(r.us[b].cpu(), o["us"][b]) < TOL
        assert _rel(r.xs_xi[b].cpu(), o["xs_xi"][b]) < TOL


def torch_isfinite(r):
    import torch
    return torch.isfinite(r.us).all() and torch.isfinite(r.xs_xi).all() and torch.isfinite(r.J_hist).all()
