"""The fused rollout + linearisation launch (k_rollout_lin, TOLG_SCHED_AUTO) against the split schedule
(separate k_rollout / k_linearize launches) and the CPU oracle.

Both schedules run the same device functions on the same values (roll_step, lin_knot); the compiler may
contract multiply-adds differently in the two kernels they are inlined into, so agreement is to rounding
(amplified by a few iterations), not bitwise.  The oracle comparison pins the pair to the reference
algorithm (tolerances as in test_gpu_parity)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bridge as ob  # noqa: E402  (test infrastructure)
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, workloads  # noqa: E402


def _oracle_problem(p):
    return ob.OracleProblem(p.kind, p.J, p.dt, p.Q, p.R, p.P, p.q_ref, p.xi_ref)


def _rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _both(prob, x0_q, x0_xi, us0, K, **kw):
    B = x0_q.shape[0]
    solver = BatchedTrackingILQR(prob, B)
    ra = solver.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=K, schedule="auto", **kw)
    keep = {k: getattr(ra, k).clone() for k in ("xs_q", "xs_xi", "us", "J_hist", "grad_hist", "defect_hist", "mu_hist",
                                                "iters", "status", "converged")}
    rs = solver.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=K, schedule="split", **kw)
    torch.cuda.synchronize()
    return keep, rs


_TOL = {"J_hist": 1e-11, "xs_q": 1e-9, "xs_xi": 1e-9, "us": 1e-9, "mu_hist": 0.0}


def _assert_same(keep, rs):
    for k, v in keep.items():
        w = getattr(rs, k)
        if not v.dtype.is_floating_point:
            assert torch.equal(v, w), k
            continue
        assert torch.equal(torch.isnan(v), torch.isnan(w)), k  # untouched history entries are NaN on both sides
        a = torch.nan_to_num(v, nan=0.0).cpu().numpy(); b = torch.nan_to_num(w, nan=0.0).cpu().numpy()
        if k in ("grad_hist", "defect_hist"):  # rounding-level quantities once converged: absolute floor
            assert np.abs(a - b).max() <= 1e-9 * np.abs(b).max() + 1e-11, (k, np.abs(a - b).max())
        else:
            assert _rel(a, b) <= _TOL[k], (k, _rel(a, b))


@pytest.mark.parametrize("B,N", [(37, 45), (16, 8), (3, 3), (64, 130), (5, 1), (20, 2)])
def test_fused_equals_split_se3(B, N):
    """Ragged shapes: B not a multiple of the 16 trajectories of a workgroup (nor of 4), N not a multiple
    of the four knots of a helper pass, N below / above the 24-knot LDS ring (slot reuse)."""
    prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N)
    keep, rs = _both(prob, x0_q, x0_xi, us0, 6, tol_grad_norm=0.0, tol_d_norm=0.0)
    _assert_same(keep, rs)
    assert (keep["status"] == 0).all() and (keep["iters"] == 6).all()


@pytest.mark.parametrize("B,N", [(8, 400), (4, 955)])
def test_fused_long_horizons(B, N):
    """The fused launch has no horizon limit (round 2 fell back to the split schedule beyond N = 313: its cost table
    lived in LDS).  N = 400 is path_se3_spiral_static_velocity's horizon, N = 955 the reference's HEAD benchmark
    problem (benchmark_SE3_tracking.py:49-58).  The auto schedule must really take the fused launch: no separate
    linearisation launch is timed."""
    prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N)
    keep, rs = _both(prob, x0_q, x0_xi, us0, 4, tol_grad_norm=0.0, tol_d_norm=0.0)
    _assert_same(keep, rs)
    assert (keep["status"] == 0).all() and (keep["iters"] == 4).all()
    solver = BatchedTrackingILQR(prob, B)
    dev = torch.device("cuda")
    solver.solve_begin(torch.as_tensor(x0_q, device=dev), torch.as_tensor(x0_xi, device=dev), torch.as_tensor(us0, device=dev),
                       mode="ms", n_iterations=3, tol_grad_norm=0.0, tol_d_norm=0.0)
    solver.enable_timing(True)
    solver.solve_iterate(3)
    torch.cuda.synchronize()
    ms_b, ms_r, ms_l, n_b = solver.kernel_time(reset=True)
    solver.enable_timing(False)
    solver.solve_end()
    assert n_b == 3 and ms_r > 0.0 and ms_l == 0.0


def test_fused_equals_split_drone_and_oracle():
    """m = 4, gravity block (the optional record field), against the oracle as well."""
    B, N, K = 9, 70, 8
    prob, x0_q, x0_xi, us0 = workloads.drone_tracking(B, N=N)
    keep, rs = _both(prob, x0_q, x0_xi, us0, K, tol_grad_norm=0.0, tol_d_norm=0.0)
    _assert_same(keep, rs)
    o = ob.fit_batch(_oracle_problem(prob), x0_q, x0_xi, us0, mode="ms", max_iter=K)
    assert _rel(keep["J_hist"].cpu(), o["J_hist"]) < 1e-9
    assert _rel(keep["us"].cpu(), o["us"]) < 1e-6
    assert _rel(keep["xs_q"].cpu(), o["xs_q"]) < 1e-6
    assert _rel(keep["xs_xi"].cpu(), o["xs_xi"]) < 1e-6


def test_fused_with_convergence_masks():
    """Trajectories converge at different iterations: finished ones (whole workgroups and single members of a
    workgroup) must be left untouched by later fused launches."""
    B = 40
    prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=60, R_scale=1e-3)
    keep, rs = _both(prob, x0_q, x0_xi, us0, 60, tol_grad_norm=1e-7)
    _assert_same(keep, rs)
    assert (keep["converged"] == 1).all()
    assert len(set(keep["iters"].cpu().tolist())) > 1
    o = ob.fit_batch(_oracle_problem(prob), x0_q[:6], x0_xi[:6], us0[:6], mode="ms", max_iter=60, tol_grad=1e-7,
                     tol_defect=1e-6)
    np.testing.assert_array_equal(keep["iters"][:6].cpu().numpy(), o["iters"])
    assert _rel(keep["us"][:6].cpu(), o["us"]) < 1e-6


def test_fused_with_al_terms():
    """Augmented-Lagrangian inner solve through the fused launch (the l_uu record field exists only here)."""
    B, N = 5, 40
    prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N, R_scale=1e-3)
    dev = torch.device("cuda")
    res = {}
    for sched in ("auto", "split"):
        solver = BatchedTrackingILQR(prob, B)
        lam = torch.full((B, N, 12), 0.3, dtype=torch.float64, device=dev)
        imu = torch.full((B, N, 12), 2.0, dtype=torch.float64, device=dev)
        solver.set_al(-0.5 * np.ones(6), 0.5 * np.ones(6), lam, imu)
        r = solver.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=5, tol_grad_norm=0.0, tol_d_norm=0.0,
                             schedule=sched)
        res[sched] = (r.us.clone(), r.J_hist.clone())
        solver.set_al(None)
    assert _rel(res["auto"][0].cpu(), res["split"][0].cpu()) < 1e-9
    assert _rel(res["auto"][1].cpu(), res["split"][1].cpu()) < 1e-11


def test_fused_so3_embedding():
    """SO3Dynamics in the SE(3) containers (TOLG_DYN_SO3) through the fused launch."""
    B, N, K = 7, 60, 6
    prob, x0_q, x0_xi, us0 = workloads.so3_tracking(B, N=N)
    keep, rs = _both(prob, x0_q, x0_xi, us0, K, tol_grad_norm=0.0, tol_d_norm=0.0)
    _assert_same(keep, rs)
    o = ob.fit_batch(_oracle_problem(prob), x0_q, x0_xi, us0, mode="ms", max_iter=K)
    assert _rel(keep["J_hist"].cpu(), o["J_hist"]) < 1e-9
    assert _rel(keep["us"].cpu(), o["us"]) < 1e-6
