"""Pendulum3dDyanmics (reference traoptlibrary/traopt_dynamics.py:421-626).

Parity unpinned against the reference: its only artefact for this model is
results_pendulum_swingup_benchmark.pkl, which no non-executing loader reads.  What is checked instead:
the oracle's f against an independent NumPy reading of fd_euler, its f_x / f_u against central finite
differences of f (the reference's Jacobians for this model are the exact ones, no quirk), and the HIP path
against the oracle (knot-wise, one linearise + backward pass, full MS line-search and SS solves)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from oracle import bridge as ob
from trajectory_optimization_matrix_lie_groups_amd import workloads

J3 = np.diag([0.5, 0.7, 0.9])
MASS, LENGTH, DT = 1.0, 0.5, 0.025
Q6 = np.diag([10.0, 10, 10, 1, 1, 1])


def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def _oracle(N=10, seed=0):
    rng = np.random.default_rng(seed)
    R_ref = np.stack([Rot.from_rotvec(rng.normal(size=3)).as_matrix() for _ in range(N + 1)])
    w_ref = rng.normal(size=(N + 1, 3))
    return ob.embed_pendulum_problem(J3, MASS, LENGTH, DT, Q6, np.eye(3) * 1e-2, 10 * Q6, R_ref, w_ref)


def test_oracle_f_matches_numpy_reading_of_fd_euler():
    op = _oracle()
    rng = np.random.default_rng(1)
    for _ in range(5):
        R = Rot.from_rotvec(rng.normal(size=3)).as_matrix(); w = rng.normal(size=3) * 2; u = rng.normal(size=3) * 3
        q, xi = ob.embed_so3_state(R, w)
        qn, xin = ob.f(op, q, xi, np.r_[u, 0, 0, 0])
        # traopt_dynamics.py:531-552, literally
        down = np.array([0, 0, -1.0]); rho = LENGTH / 2 * down
        g_term = _skew(MASS * 9.8 * rho) @ (R.T @ down)
        Mt = _skew(MASS * rho) @ (R.T @ u)
        w_next = w + np.linalg.inv(J3) @ (_skew(w).T @ J3 @ w + g_term + Mt) * DT
        R_next = R @ Rot.from_rotvec(w * DT).as_matrix()
        assert np.abs(qn[:3, :3] - R_next).max() < 1e-14
        assert np.abs(xin[:3] - w_next).max() < 1e-14
        assert np.all(xin[3:] == 0) and np.all(qn[:3, 3] == 0)


def test_oracle_jacobians_match_finite_differences():
    op = _oracle()
    rng = np.random.default_rng(2)
    rows = [0, 1, 2, 6, 7, 8]
    for _ in range(3):
        q, xi = ob.embed_so3_state(Rot.from_rotvec(rng.normal(size=3)).as_matrix(), rng.normal(size=3) * 2)
        u = np.r_[rng.normal(size=3) * 3, 0, 0, 0]
        Fx, Fu = ob.fx_fu(op, q, xi, u)
        qn, xin = ob.f(op, q, xi, u)

        def diff(qa, xa):
            return np.r_[ob.rminus(qa, qn), xa - xin]

        eps = 1e-6
        Fx_fd = np.zeros((12, 12)); Fu_fd = np.zeros((12, 6))
        for k in rows:
            d = np.zeros(12); d[k] = eps
            a = ob.f(op, q @ ob.se3_exp(d[:6]), xi + d[6:], u); b = ob.f(op, q @ ob.se3_exp(-d[:6]), xi - d[6:], u)
            Fx_fd[:, k] = (diff(*a) - diff(*b)) / (2 * eps)
        for k in range(3):
            du = np.zeros(6); du[k] = eps
            Fu_fd[:, k] = (diff(*ob.f(op, q, xi, u + du)) - diff(*ob.f(op, q, xi, u - du))) / (2 * eps)
        sub = np.ix_(rows, rows)
        assert np.abs(Fx[sub] - Fx_fd[sub]).max() < 1e-8
        assert np.abs(Fu[rows][:, :3] - Fu_fd[rows][:, :3]).max() < 1e-8
        # closed forms: F_u = J^-1 skew(m rho) R^T dt; lower-left = J^-1 skew(m rho) skew(R^T (g e + u)) dt
        R = q[:3, :3]; rho = np.array([0, 0, -LENGTH / 2])
        assert np.abs(Fu[6:9, :3] - np.linalg.inv(J3) @ _skew(MASS * rho) @ R.T * DT).max() < 1e-15
        wv = R.T @ (np.array([0, 0, -9.8]) + u[:3])
        assert np.abs(Fx[6:9, :3] - np.linalg.inv(J3) @ _skew(MASS * rho) @ _skew(wv) * DT).max() < 1e-13


def test_oracle_swingup_converges():
    prob, x0_q, x0_xi, us0 = workloads.pendulum_swingup(1)
    op = ob.embed_pendulum_problem(J3, MASS, LENGTH, prob.dt, Q6, np.eye(3) * 1e-2, 10 * Q6, prob.q_ref[:, :3, :3],
                                   prob.xi_ref[:, :3])
    o = ob.fit(op, x0_q[0], x0_xi[0], us0[0], mode="ms", max_iter=60, line_search=True, rollout="nonlinear")
    J = o["J_hist"][: o["n_iters"]]
    assert o["n_iters"] >= 5 and J[-1] < J[0] and np.isfinite(J).all()
    assert o["defect_hist"][o["n_iters"]] < 1e-6  # the multiple-shooting gaps close


def _oracle_of(prob):
    return ob.embed_pendulum_problem(prob.J[:3, :3], prob.pend_mass, prob.pend_length, prob.dt,
                                     np.block([[prob.Q[:3, :3], np.zeros((3, 3))], [np.zeros((3, 3)), prob.Q[6:9, 6:9]]]),
                                     prob.R[:3, :3],
                                     np.block([[prob.P[:3, :3], np.zeros((3, 3))], [np.zeros((3, 3)), prob.P[6:9, 6:9]]]),
                                     prob.q_ref[:, :3, :3], prob.xi_ref[:, :3])


def _rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.gpu
def test_gpu_pendulum_linearize_backward_matches_oracle():
    from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR
    B = 5
    prob, x0_q, x0_xi, _ = workloads.pendulum_swingup(B)
    N = prob.N
    rng = np.random.default_rng(3)
    xs_q = np.tile(np.eye(4), (B, N + 1, 1, 1)); xs_xi = np.zeros((B, N + 1, 6)); us = np.zeros((B, N, 6))
    for b in range(B):
        for i in range(N + 1):
            xs_q[b, i, :3, :3] = prob.q_ref[i, :3, :3] @ Rot.from_rotvec(rng.normal(size=3) * 0.4).as_matrix()
            xs_xi[b, i, :3] = prob.xi_ref[i, :3] + rng.normal(size=3)
        us[b, :, :3] = rng.normal(size=(N, 3)) * 3
    solver = BatchedTrackingILQR(prob, B)
    op = _oracle_of(prob)
    for ms in (True, False):
        r = solver.linearize_backward(xs_q, xs_xi, us, ms=ms)
        for b in range(B):
            o = ob.lin_backward(op, xs_q[b], xs_xi[b], us[b], ms=ms)
            assert _rel(r["Fx"][b].cpu(), o["Fx"]) < 1e-12
            assert np.abs(r["d"][b].cpu().numpy() - o["d"]).max() < 1e-11 * max(1.0, np.abs(o["d"]).max())
            assert _rel(r["lx"][b].cpu(), o["Lx"]) < 1e-11
            assert float(r["J"][b]) == pytest.approx(o["J"], rel=1e-12)
            assert _rel(r["K"][b].cpu(), o["K"]) < 1e-8
            assert _rel(r["k"][b].cpu(), o["k"]) < 1e-8
            assert float(r["grad"][b]) == pytest.approx(o["grad"], rel=1e-9)


@pytest.mark.gpu
def test_gpu_pendulum_knot_quantities_match_oracle():
    from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR
    prob, _, _, _ = workloads.pendulum_swingup(1)
    op = _oracle_of(prob)
    solver = BatchedTrackingILQR(prob, 8)
    rng = np.random.default_rng(5)
    n, i = 8, 17
    xq = np.tile(np.eye(4), (n, 1, 1)); xxi = np.zeros((n, 6)); u = np.zeros((n, 6))
    for k in range(n):
        xq[k, :3, :3] = Rot.from_rotvec(rng.normal(size=3)).as_matrix()
        xxi[k, :3] = rng.normal(size=3) * 2
        u[k, :3] = rng.normal(size=3) * 4
    r = solver.eval_knot(i, xq, xxi, u)
    for k in range(n):
        qn, xin = ob.f(op, xq[k], xxi[k], u[k])
        Fx, Fu = ob.fx_fu(op, xq[k], xxi[k], u[k])
        assert np.abs(r["f_q"][k].cpu().numpy() - qn).max() < 1e-14
        assert np.abs(r["f_xi"][k].cpu().numpy() - xin).max() < 1e-13
        assert _rel(r["Fx"][k].cpu(), Fx) < 1e-13
        assert _rel(r["Fu"][k].cpu(), Fu) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("spread", [0.03, 0.4])
def test_gpu_pendulum_expected_change_ring_matches_statement_form(spread):
    """The merit search's preparation for this model in its ring form (k_expected_change_ring<.., DENSE, VARB>: the velocity block
    and the knot's input matrix read from the record run) against the kernel that walks the reference's statements
    (traopt_controller.py:2730-2737, :2756-2769 over Pendulum3dDyanmics.f_x / f_u, traopt_dynamics.py:566-609), on open trajectories
    with defects of size `spread`; what the ring form hands back the statement form fills in."""
    import torch
    from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR
    B = 9
    prob, *_ = workloads.pendulum_swingup(B)
    N = prob.N
    rng = np.random.default_rng(11)
    xs_q = np.tile(np.eye(4), (B, N + 1, 1, 1)); xs_xi = np.zeros((B, N + 1, 6)); us = np.zeros((B, N, 6))
    for b in range(B):
        for i in range(N + 1):
            xs_q[b, i, :3, :3] = prob.q_ref[i, :3, :3] @ Rot.from_rotvec(rng.normal(size=3) * spread).as_matrix()
            xs_xi[b, i, :3] = prob.xi_ref[i, :3] + rng.normal(size=3) * spread
        us[b, :, :3] = rng.normal(size=(N, 3))
    solver = BatchedTrackingILQR(prob, B)
    solver.linearize_backward(xs_q, xs_xi, us, ms=True)
    es, _ = solver.expected_change(B, "statement")
    er, flag = solver.expected_change(B, "ring")
    er2, flag2 = solver.expected_change(B, "ring")
    ea, _ = solver.expected_change(B, "auto")
    torch.cuda.synchronize()
    es, er, er2, ea, flag, flag2 = (t.cpu().numpy() for t in (es, er, er2, ea, flag, flag2))
    np.testing.assert_array_equal(flag, flag2)
    np.testing.assert_array_equal(er, er2)
    keep = flag == 0
    assert keep.sum() >= (B // 2 if spread < 0.1 else 1)
    assert np.isfinite(es).all()
    scale = np.abs(es).max(axis=1, keepdims=True)
    assert (np.abs(er[keep] - es[keep]) / scale[keep]).max() < 1e-11
    assert np.isnan(er[~keep]).all()
    np.testing.assert_array_equal(ea[~keep], es[~keep])
    np.testing.assert_array_equal(ea[keep], er[keep])


@pytest.mark.gpu
def test_gpu_pendulum_merit_search_ring_and_statement_schedules_agree():
    """A merit-search solve with the ring form (schedule auto) and with the statement form alone (split): same step sizes, costs
    to rounding, and the oracle's."""
    from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR
    B, K = 7, 10
    prob, x0_q, x0_xi, us0 = workloads.pendulum_swingup(B, xi0_scale=1.0)
    res = {}
    for sched in ("auto", "split"):
        r = BatchedTrackingILQR(prob, B).fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0,
                                                   line_search=True, rollout="nonlinear", schedule=sched)
        res[sched] = (r.J_hist.cpu().numpy(), r.alpha_hist.cpu().numpy(), r.iters.cpu().numpy())
    (Ja, Aa, ia), (Js, As, is_) = res["auto"], res["split"]
    np.testing.assert_array_equal(ia, is_)
    o = ob.fit_batch(_oracle_of(prob), x0_q, x0_xi, us0, mode="ms", max_iter=K, line_search=True, rollout="nonlinear")
    for b in range(B):
        n = ia[b]
        assert _rel(Ja[b, :n], Js[b, :n]) < 1e-10
        np.testing.assert_allclose(Aa[b, : n - 1], As[b, : n - 1], rtol=1e-14)
        m = min(n, o["iters"][b])
        assert _rel(Ja[b, :m], o["J_hist"][b, :m]) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("mode,line_search,rollout", [("ms", True, "nonlinear"), ("ms", False, "nonlinear"),
                                                      ("ss", False, "nonlinear"), ("ms", False, "linear"),
                                                      ("ms", True, "linear"), ("ss", False, "linear")])  # (the affine path: every candidate from one sweep)
def test_gpu_pendulum_fit_matches_oracle(mode, line_search, rollout):
    from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR
    B, K = 6, 12
    prob, x0_q, x0_xi, us0 = workloads.pendulum_swingup(B, xi0_scale=1.0)
    solver = BatchedTrackingILQR(prob, B)
    r = solver.fit_batch(x0_q, x0_xi, us0, mode=mode, n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0,
                         line_search=line_search, rollout=rollout)
    o = ob.fit_batch(_oracle_of(prob), x0_q, x0_xi, us0, mode=mode, max_iter=K, line_search=line_search, rollout=rollout)
    it = r.iters.cpu().numpy()
    np.testing.assert_array_equal(it, o["iters"])
    Jg = r.J_hist.cpu().numpy()
    for b in range(B):
        n = it[b]
        assert _rel(Jg[b, :n], o["J_hist"][b, :n]) < 1e-8
    assert _rel(r.us.cpu().numpy()[:, :, :3], o["us"][:, :, :3]) < 1e-6
    assert np.abs(r.us.cpu().numpy()[:, :, 3:]).max() == 0.0


@pytest.mark.gpu
def test_gpu_pendulum_mirror_plugin_and_controller():
    """The reference-named classes (main_pendulum3d_ddp_tracking_exact_ms.py:93-127) on the HIP path."""
    from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary.traopt_controller import iLQR_Tracking_SO3_MS
    from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary.traopt_cost import SO3TrackingQuadraticGaussNewtonCost
    from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary.traopt_dynamics import Pendulum3dDyanmics
    from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary.traopt_utilis import SO3, SO3Tangent
    R_ref, w_ref, dt = workloads.load_reference("pendulum_swingup_n80")
    N = R_ref.shape[0] - 1
    dyn = Pendulum3dDyanmics(J3, 1, 0.5, dt, hessians=False)
    assert (dyn.state_size, dyn.action_size, dyn.m, dyn.l, dyn.g) == (6, 3, 1, 0.5, 9.8)
    cost = SO3TrackingQuadraticGaussNewtonCost(Q6, np.eye(3) * 1e-2, Q6 * 10, R_ref, w_ref)
    R0 = Rot.from_euler("xy", [10.0, 45.0], degrees=True).as_matrix()
    x0 = [SO3.from_matrix(R0), SO3Tangent(np.array([1.0, 1.0, 0.0]))]
    op = ob.embed_pendulum_problem(J3, 1.0, 0.5, dt, Q6, np.eye(3) * 1e-2, 10 * Q6, R_ref, w_ref)
    q0, xi0 = ob.embed_so3_state(R0, [1.0, 1.0, 0.0])
    # plugin methods
    u = np.array([0.3, -1.2, 2.0])
    Fx, Fu = ob.fx_fu(op, q0, xi0, np.r_[u, 0, 0, 0])
    idx = [0, 1, 2, 6, 7, 8]
    assert np.abs(dyn.f_x(x0, u, 0) - Fx[np.ix_(idx, idx)]).max() < 1e-13
    assert np.abs(dyn.f_u(x0, u, 0) - Fu[np.ix_(idx, [0, 1, 2])]).max() < 1e-13
    qn, xin = ob.f(op, q0, xi0, np.r_[u, 0, 0, 0])
    fq, fxi = dyn.f(x0, u, 0)
    assert np.abs(fq.rotation() - qn[:3, :3]).max() < 1e-14 and np.abs(fxi.coeffs() - xin[:3]).max() < 1e-13
    # controller
    calls = []
    ilqr = iLQR_Tracking_SO3_MS(dyn, cost, N, R_ref, w_ref, hessians=False, line_search=True, rollout="nonlinear")
    xs, us, J_hist, xs_hist, us_hist, grad_hist, defect_hist = ilqr.fit(
        x0, np.zeros((N, 3)), n_iterations=15, on_iteration=lambda *a: calls.append(a))
    o = ob.fit(op, q0, xi0, np.zeros((N, 6)), mode="ms", max_iter=15, line_search=True, rollout="nonlinear")
    n = o["n_iters"]
    assert len(calls) == n
    Jc = np.array([c[3] for c in calls])
    assert _rel(Jc, o["J_hist"][:n]) < 1e-8
    assert _rel(us, o["us"][:, :3]) < 1e-6
