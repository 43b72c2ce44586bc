"""The merit search's linear alpha = 1 rollout in its two forms (GPU, -m gpu): k_expected_change_ring -- the affine
recursion in the backward sweep's lane map, inputs through an LDS ring -- against k_expected_change, which walks the
reference's statements (traopt_controller.py:2550-2557, :2730-2737, :2756-2788), and both against the oracle.
schedule="split" selects the statement-by-statement kernel alone; "auto" the ring form with the hand-back behind it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bridge as ob  # noqa: E402  (test infrastructure)
from trajectory_optimization_matrix_lie_groups_amd import BatchedTrackingILQR, TrackingProblem, workloads  # noqa: E402


def _oracle_problem(p: TrackingProblem):
    return ob.OracleProblem(p.kind, p.J, p.dt, p.Q, p.R, p.P, p.q_ref, p.xi_ref)


def _rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _dense(prob):
    """The same problem with full inertia blocks (a rotated body frame's inertia): the backward sweep and the ring kernel then read
    I + H dt from the record (Params::fA22 >= 0) instead of rebuilding it from the twist."""
    A = np.array([[0.10, -0.05, 0.02], [0.03, 0.12, -0.04], [-0.02, 0.06, 0.09]])
    Jd = np.array(prob.J, dtype=float).copy()
    Jd[:3, :3] += A @ A.T
    if prob.kind == "se3":
        Jd[3:, 3:] += 0.5 * (A @ A.T)
    return TrackingProblem(prob.kind, Jd, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)


def _problem(kind, B, N):
    if kind.endswith("_dense"):
        prob, x0_q, x0_xi, us0 = _problem(kind[:-6], B, N)
        return _dense(prob), x0_q, x0_xi, us0
    if kind == "se3":
        return workloads.se3_tracking(B, N=N, R_scale=1e-3)
    if kind == "drone":
        return workloads.drone_tracking(B, N=N, R_scale=1e-3)
    prob, x0_q, x0_xi, us0 = workloads.so3_tracking(B, N=N)
    return prob, x0_q, x0_xi, us0


def _random_traj(prob, B, seed, spread):
    rng = np.random.default_rng(seed)
    N, m = prob.N, prob.m
    xs_q = np.empty((B, N + 1, 4, 4)); xs_xi = np.empty((B, N + 1, 6)); us = rng.normal(size=(B, N, m)) * 0.3
    for b in range(B):
        for i in range(N + 1):
            xs_q[b, i] = prob.q_ref[i] @ ob.se3_exp(rng.normal(size=6) * spread * (1 if prob.kind != "so3" else np.r_[1, 1, 1, 0, 0, 0]))
            xs_xi[b, i] = prob.xi_ref[i] + rng.normal(size=6) * spread * (1 if prob.kind != "so3" else np.r_[1, 1, 1, 0, 0, 0])
    if prob.kind == "so3":
        us[:, :, 3:] = 0
    return xs_q, xs_xi, us


@pytest.mark.parametrize("kind,B,N", [("se3", 7, 33), ("se3", 64, 200), ("drone", 5, 60), ("drone", 12, 150), ("so3", 4, 40),
                                      ("se3", 1, 1), ("se3", 3, 2), ("drone", 2, 3), ("se3", 5, 5), ("se3", 17, 6),  # horizons shorter than the ring
                                      ("se3_dense", 7, 33), ("se3_dense", 21, 120), ("drone_dense", 6, 50), ("se3_dense", 3, 2)])
@pytest.mark.parametrize("spread", [0.02, 0.15])
def test_ring_kernel_matches_statement_kernel_on_random_trajectories(kind, B, N, spread):
    """The two kernels on the same records and gains (open trajectories with defects of size `spread`): first- and
    second-order terms agree to rounding wherever the ring form keeps a trajectory; repeated launches are bitwise
    reproducible; what it hands back, the statement kernel behind it fills in.  (The entry point poisons the LDS of
    every CU with NaNs first: the statement kernel once filled its constant table AFTER inactive quads had left, and
    passed wherever the previous launch had left the same table behind.)"""
    prob, *_ = _problem(kind, 1, N)
    xs_q, xs_xi, us = _random_traj(prob, B, seed=3 + N, spread=spread)
    if B >= 5:  # two trajectories with rotation defects near pi: candidates for the hand-back
        wild, _, _ = _random_traj(prob, B, seed=4 + N, spread=1.6)
        xs_q[1], xs_q[4] = wild[1], wild[4]
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
    assert keep.sum() >= B // 2
    assert np.isfinite(es).all()
    scale = np.abs(es).max(axis=1, keepdims=True)
    assert (np.abs(er[keep] - es[keep]) / scale[keep]).max() < 1e-11
    assert np.isnan(er[~keep]).all()
    np.testing.assert_array_equal(ea[~keep], es[~keep])
    np.testing.assert_array_equal(ea[keep], er[keep])


def _fit(solver, x0_q, x0_xi, us0, K, schedule):
    r = solver.fit_batch(x0_q, x0_xi, us0, mode="ms", n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0,
                         line_search=True, schedule=schedule)
    torch.cuda.synchronize()
    return r


def _same_search(it_a, J_a, A_a, it_b, J_b, A_b, tol, what):
    """Two runs of the same searches: costs (and, if given, accepted step sizes) agree over the iterations both made;
    they end together -- or one trial apart where the search has converged to rounding level (J_new < J_opt, resp. the
    Armijo test, with both sides equal to 13 digits is a coin flip; tests/test_gpu_matrix.py has the same allowance).
    Returns the trajectories that ended together."""
    same = []
    for b in range(len(it_a)):
        n = min(int(it_a[b]), int(it_b[b]))
        if n:
            assert _rel(J_a[b, :n], J_b[b, :n]) < tol, what
            if A_a is not None:
                np.testing.assert_allclose(A_a[b, : n - 1], A_b[b, : n - 1], rtol=1e-14, err_msg=what)
        if it_a[b] == it_b[b]:
            same.append(b)
            continue
        longer = J_a[b, : it_a[b]] if it_a[b] > it_b[b] else J_b[b, : it_b[b]]
        tail = longer[max(n - 1, 0):]
        assert abs(int(it_a[b]) - int(it_b[b])) <= 1 and np.ptp(tail) <= 1e-11 * abs(tail[0]), what
    assert same, what
    return same


@pytest.mark.parametrize("kind,B,N", [("se3", 6, 50), ("se3", 9, 37), ("drone", 5, 60), ("so3", 4, 40), ("se3_dense", 7, 45), ("drone_dense", 5, 40)])
def test_ring_form_matches_statement_form_and_oracle(kind, B, N):
    """Same accepted step sizes, same costs (the two kernels differ by Exp/Log round trips and the order of two sums);
    B not a multiple of 4 exercises the padded group, N not a multiple of 4 the ring's tail steps."""
    K = 8
    prob, x0_q, x0_xi, us0 = _problem(kind, B, N)
    solver = BatchedTrackingILQR(prob, B)
    ra = _fit(solver, x0_q, x0_xi, us0, K, "auto")
    Ja, Aa, ia, ua = ra.J_hist.cpu().numpy().copy(), ra.alpha_hist.cpu().numpy().copy(), ra.iters.cpu().numpy().copy(), ra.us.cpu().numpy().copy()
    rs = _fit(solver, x0_q, x0_xi, us0, K, "split")
    Js, As, is_, us_ = rs.J_hist.cpu().numpy(), rs.alpha_hist.cpu().numpy(), rs.iters.cpu().numpy(), rs.us.cpu().numpy()
    o = ob.fit_batch(_oracle_problem(prob), x0_q, x0_xi, us0, mode="ms", max_iter=K, line_search=True)
    same = _same_search(ia, Ja, Aa, is_, Js, As, 1e-10, "ring form vs statement form")
    assert _rel(ua[same], us_[same]) < 1e-8
    _same_search(is_, Js, None, o["iters"], o["J_hist"], None, 1e-8, "statement form vs oracle")
    same = _same_search(ia, Ja, None, o["iters"], o["J_hist"], None, 1e-8, "ring form vs oracle")
    assert _rel(ua[same], o["us"][same]) < 1e-6


def test_large_rotation_deviation_is_handed_back():
    """An initial pose more than 3 rad away from the reference's first knots: the linear rollout's rotation deviation
    leaves the range in which Log(Exp(v)) = v, the ring kernel flags the trajectory and the statement-by-statement
    kernel behind it produces the result -- the same one as without the ring kernel, and the oracle's."""
    B, N, K = 6, 40, 6
    prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N, R_scale=1e-3)
    # trajectories 1, 3, 4: the reference's first pose turned by 3.1 rad about some axis; the others stay as they are
    rng = np.random.default_rng(5)
    for b in (1, 3, 4):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        x0_q[b] = prob.q_ref[0] @ ob.se3_exp(np.concatenate([3.1 * ax, rng.normal(size=3) * 0.2]))
    solver = BatchedTrackingILQR(prob, B)
    ra = _fit(solver, x0_q, x0_xi, us0, K, "auto")
    Ja, ia, ua = ra.J_hist.cpu().numpy().copy(), ra.iters.cpu().numpy().copy(), ra.us.cpu().numpy().copy()
    Aa = ra.alpha_hist.cpu().numpy().copy()
    rs = _fit(solver, x0_q, x0_xi, us0, K, "split")
    o = ob.fit_batch(_oracle_problem(prob), x0_q, x0_xi, us0, mode="ms", max_iter=K, line_search=True)
    _same_search(ia, Ja, Aa, rs.iters.cpu().numpy(), rs.J_hist.cpu().numpy(), rs.alpha_hist.cpu().numpy(), 1e-10,
                 "ring form + hand-back vs statement form")
    same = _same_search(ia, Ja, None, o["iters"], o["J_hist"], None, 1e-8, "ring form + hand-back vs oracle")
    assert _rel(ua[same], o["us"][same]) < 1e-6


@pytest.mark.parametrize("mode,line_search", [("ss", False), ("ms", True)])
def test_wide_stage_forms_agree_at_the_metric_size(mode, line_search):
    """A wide line-search stage runs as quad rollouts over the compacted list of undecided trajectories while that list is
    short, as one thread per (trajectory, alpha) when it is long (ls_quad_form, decided on the device).  Only a large
    batch reaches the second form: the same 12 trajectories solved inside the 4096 x 200 batch of the metric and in a
    batch of their own (short lists, the form every oracle comparison of the suite exercises) must agree -- and in the
    large batch the list must indeed have been long (SS: a third of the trajectories backtrack in the early iterations)."""
    B, N, K = 4096, 200, 5
    prob, x0_q, x0_xi, us0 = workloads.se3_tracking(B, N=N)
    pick = np.array([0, 1, 2, 3, 500, 501, 1023, 2048, 2049, 3000, 4094, 4095])
    big = BatchedTrackingILQR(prob, B)
    rb = big.fit_batch(x0_q, x0_xi, us0, mode=mode, n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0, line_search=line_search)
    torch.cuda.synchronize()
    Ab, Jb, ib = rb.alpha_hist.cpu().numpy(), rb.J_hist.cpu().numpy(), rb.iters.cpu().numpy()
    small = BatchedTrackingILQR(prob, len(pick))
    rs = small.fit_batch(x0_q[pick], x0_xi[pick], us0[pick], mode=mode, n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0,
                         line_search=line_search)
    torch.cuda.synchronize()
    _same_search(ib[pick], Jb[pick], Ab[pick], rs.iters.cpu().numpy(), rs.J_hist.cpu().numpy(), rs.alpha_hist.cpu().numpy(),
                 1e-9, "large batch vs batch of twelve")
    if mode == "ss":  # the thread form's turn needs more than 20 000 / 12 undecided trajectories in some iteration
        backtracked = ((Ab[:, :K] < 1.0) & (np.arange(K)[None, :] < ib[:, None])).sum(axis=0)
        assert backtracked.max() > 20000 // 12, backtracked


@pytest.mark.parametrize("kind", ["se3", "drone"])
def test_al_terms_in_both_forms_and_in_the_staged_search(kind):
    """Augmented-Lagrangian solves carry one more record field (the l_uu diagonal) and a cost with multiplier terms: the
    ring kernel reads the first, k_ls_eval evaluates the second.  Kernel against kernel on a random trajectory, then a
    merit search with the ring form against one with the statement form (and single shooting, which has only the
    staged evaluation to differ in, against itself through both schedules)."""
    B, N, K = 6, 45, 5
    prob, x0_q, x0_xi, us0 = _problem(kind, B, N)
    m = prob.m
    dev = torch.device("cuda")
    rng = np.random.default_rng(2)
    lam = torch.tensor(rng.uniform(0.0, 0.5, size=(B, N, 2 * m)), dtype=torch.float64, device=dev)
    imu = torch.tensor(rng.uniform(0.5, 3.0, size=(B, N, 2 * m)), dtype=torch.float64, device=dev)
    solver = BatchedTrackingILQR(prob, B)
    solver.set_al(-0.4 * np.ones(m), 0.4 * np.ones(m), lam, imu)
    xs_q, xs_xi, us = _random_traj(prob, B, seed=9, spread=0.05)
    solver.linearize_backward(xs_q, xs_xi, us, ms=True)
    es, _ = solver.expected_change(B, "statement")
    er, flag = solver.expected_change(B, "ring")
    torch.cuda.synchronize()
    es, er, flag = es.cpu().numpy(), er.cpu().numpy(), flag.cpu().numpy()
    assert (flag == 0).all()
    assert (np.abs(er - es) / np.abs(es).max(axis=1, keepdims=True)).max() < 1e-11
    res = {}
    for mode, ls in (("ms", True), ("ss", False)):
        for sched in ("auto", "split"):
            r = solver.fit_batch(x0_q, x0_xi, us0, mode=mode, n_iterations=K, tol_grad_norm=0.0, tol_d_norm=0.0,
                                 line_search=ls, schedule=sched)
            torch.cuda.synchronize()
            res[mode, sched] = (r.iters.cpu().numpy().copy(), r.J_hist.cpu().numpy().copy(), r.alpha_hist.cpu().numpy().copy())
        a, s_ = res[mode, "auto"], res[mode, "split"]
        _same_search(a[0], a[1], a[2], s_[0], s_[1], s_[2], 1e-10, "AL %s: auto vs split schedule" % mode)
    solver.set_al(None)
