"""The generic plugin path of the SE(3) controllers (traoptlibrary/_generic_lie.py): plugins outside the
closed-form set run the per-knot callback loop (SURVEY.md §8b "What calls it"; reference
traopt_controller.py:2823-2910).  Here the plugins are hand-written BaseDynamics / BaseCost classes whose
methods evaluate the CPU oracle's per-knot functions, so the whole loop can be compared with the oracle's own
solver -- on the CPU, no GPU involved."""
import warnings

import numpy as np
import pytest

from oracle import bridge as ob
from trajectory_optimization_matrix_lie_groups_amd import workloads
from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary.traopt_controller import (iLQR_Tracking_SE3,
                                                                                           iLQR_Tracking_SE3_MS)
from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary.traopt_cost import BaseCost
from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary.traopt_dynamics import BaseDynamics


class MyDynamics(BaseDynamics):
    """A user-defined plugin: rigid body on SE(3) evaluated by the oracle's per-knot functions."""

    def __init__(self, op, m):
        self._op, self._m = op, m
        self._error_state_size = 6

    state_size = property(lambda self: 12)
    action_size = property(lambda self: self._m)
    has_hessians = property(lambda self: False)

    def f(self, x, u, i):
        q, xi = ob.f(self._op, x[0], x[1], u)
        return [q, xi]

    def f_x(self, x, u, i):
        return ob.fx_fu(self._op, x[0], x[1], u)[0]

    def f_u(self, x, u, i):
        return ob.fx_fu(self._op, x[0], x[1], u)[1]

    def f_xx(self, x, u, i): raise NotImplementedError
    def f_ux(self, x, u, i): raise NotImplementedError
    def f_uu(self, x, u, i): raise NotImplementedError


class MyCost(BaseCost):
    def __init__(self, op, m):
        self._op, self._m = op, m

    def _all(self, x, u, i, terminal):
        return ob.cost(self._op, x[0], x[1], u, i, terminal)

    def l(self, x, u, i, terminal=False): return self._all(x, u, i, terminal)[0]  # noqa: E704,E741
    def l_x(self, x, u, i, terminal=False): return self._all(x, u, i, terminal)[1]  # noqa: E704
    def l_u(self, x, u, i, terminal=False): return self._all(x, u, i, terminal)[3]  # noqa: E704
    def l_xx(self, x, u, i, terminal=False): return self._all(x, u, i, terminal)[2]  # noqa: E704
    def l_ux(self, x, u, i, terminal=False): return np.zeros((self._m, 12))  # noqa: E704
    def l_uu(self, x, u, i, terminal=False): return self._all(x, u, i, terminal)[4]  # noqa: E704


@pytest.mark.parametrize("kind", ["se3", "drone"])
@pytest.mark.parametrize("mode", ["ms", "ss"])
def test_generic_loop_matches_the_oracle_solver(kind, mode):
    N, K = 20, 6
    if kind == "se3":
        prob, x0_q, x0_xi, us0 = workloads.se3_tracking(1, N=N, R_scale=1e-3)
    else:
        prob, x0_q, x0_xi, us0 = workloads.drone_tracking(1, N=N, R_scale=1e-3)
    op = ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
    dyn, cost = MyDynamics(op, prob.m), MyCost(op, prob.m)
    calls = []

    def cb(*a):
        calls.append(a)
        a[-5 if mode == "ms" else -3].append(a[3])   # J_hist is filled by the callback, as in the reference

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if mode == "ms":
            ctl = iLQR_Tracking_SE3_MS(dyn, cost, N, prob.q_ref, prob.xi_ref, rollout="nonlinear")
            xs, us, J_hist, _, _, grad_hist, defect_hist = ctl.fit([x0_q[0], x0_xi[0]], us0[0], n_iterations=K,
                                                                    tol_grad_norm=0.0, on_iteration=cb)
        else:
            ctl = iLQR_Tracking_SE3(dyn, cost, N, rollout="nonlinear")
            xs, us, J_hist, _, _, grad_hist = ctl.fit([x0_q[0], x0_xi[0]], us0[0], n_iterations=K, tol_grad_norm=0.0,
                                                      on_iteration=cb)
    o = ob.fit(op, x0_q[0], x0_xi[0], us0[0], mode=mode, max_iter=K, tol_grad=0.0)
    n = int(o["n_iters"])
    assert len(calls) == n and len(calls[0]) == (15 if mode == "ms" else 12)
    assert np.allclose(J_hist, o["J_hist"][:n], rtol=1e-9)
    assert np.abs(us - o["us"]).max() < 1e-7 * max(1.0, np.abs(o["us"]).max())
    assert np.abs(np.stack([x[0] for x in xs]) - o["xs_q"]).max() < 1e-8
    assert np.abs(np.stack([x[1] for x in xs]) - o["xs_xi"]).max() < 1e-7 * max(1.0, np.abs(o["xs_xi"]).max())
    if mode == "ms":
        assert defect_hist[0] == pytest.approx(o["defect_hist"][0], rel=1e-10)


def test_merit_search_is_fused_only():
    prob, *_ = workloads.se3_tracking(1, N=5)
    op = ob.OracleProblem(prob.kind, prob.J, prob.dt, prob.Q, prob.R, prob.P, prob.q_ref, prob.xi_ref)
    ctl = iLQR_Tracking_SE3_MS(MyDynamics(op, 6), MyCost(op, 6), 5, prob.q_ref, prob.xi_ref, line_search=True)
    with pytest.raises(NotImplementedError):
        ctl.fit([prob.q_ref[0], prob.xi_ref[0]], np.zeros((5, 6)))
