"""Controllers with the reference's interface (traoptlibrary/traopt_controller.py).

BaseController :14-40, iLQR_Tracking_SE3 :1831-2349, iLQR_Tracking_SE3_MS :2352-3136,
AL_iLQR_Tracking_SE3_MS :3139-3293.  ``fit`` keeps the reference's signature, return tuple and
callback protocol for one trajectory; ``fit_batch`` (new) solves B initial states at once.  Both
dispatch the whole iteration loop to the HIP extension; there is no per-knot Python loop and no CPU
fallback."""
import abc
import warnings

import numpy as np

from ..solver import BatchedTrackingILQR, TrackingProblem
from . import _bridge
from .traopt_cost import ALConstrainedCost, SE3TrackingQuadraticGaussNewtonCost, SO3TrackingQuadraticGaussNewtonCost
from .traopt_dynamics import DroneDynamics, Pendulum3dDyanmics, RigidBodyDynamics, SE3Dynamics, SO3Dynamics

_KIND = {SE3Dynamics: "se3", RigidBodyDynamics: "rigidbody", DroneDynamics: "drone"}
_MSG_MAXREG = "exceeded max regularization term"  # traopt_controller.py:2984
_MSG_NODESCENT = "Couldn't find descent direction, regularization and line search step exhausted"  # :2632


class BaseController():
    """Base trajectory optimizer controller (traopt_controller.py:14-40)."""

    @abc.abstractmethod
    def fit(self, x0, us_init, *args, **kwargs):
        raise NotImplementedError


def _problem_of(dynamics, cost):
    """The fused path exists for the closed-form classes only (SURVEY.md §8b 'What calls it')."""
    if type(dynamics) not in _KIND:
        raise TypeError("the MI355X path supports SE3Dynamics / RigidBodyDynamics / DroneDynamics, got %s"
                        % type(dynamics).__name__)
    base = cost.cost if isinstance(cost, ALConstrainedCost) else cost
    if type(base) is not SE3TrackingQuadraticGaussNewtonCost:
        raise TypeError("the MI355X path supports SE3TrackingQuadraticGaussNewtonCost, got %s" % type(base).__name__)
    if base.action_size != dynamics.action_size:
        raise ValueError("cost.action_size (%d) != dynamics.action_size (%d)" % (base.action_size, dynamics.action_size))
    return TrackingProblem(_KIND[type(dynamics)], dynamics.J, dynamics.dt, base.Q, base.R, base.P, base._q_ref_mats,
                           base._xi_ref)


def _xs_list(xs_q, xs_xi):
    return [[xs_q[i].copy(), xs_xi[i].copy()] for i in range(xs_q.shape[0])]


class _FusedController(BaseController):
    _mode = "ms"
    _append_grad_on_convergence = False  # iLQR_Tracking_SO3_MS does (traopt_controller.py:1219)

    def _common_init(self, dynamics, cost, N, max_reg, hessians, rollout, debug):
        self.dynamics = dynamics
        self.cost = cost
        self.N = N
        self._use_hessians = hessians and dynamics.has_hessians
        if hessians and not dynamics.has_hessians:
            warnings.warn("hessians requested but are unavailable in dynamics")  # traopt_controller.py:2385
        if self._use_hessians:
            raise NotImplementedError("second-order dynamics terms do not exist for the exact SE3 models")
        self._mu = 1.0
        self._mu_min = 1e-6
        self._mu_max = max_reg
        self._delta_0 = 2.0
        self._delta = self._delta_0
        self._action_size = dynamics.action_size
        self._state_size = dynamics.state_size
        self._error_state_size = dynamics._error_state_size
        self._rollout_mode = rollout
        self._debug = debug
        self._k = np.zeros((N, self._action_size))
        self._K = np.zeros((N, self._action_size, self._state_size))
        self._solver = None
        self._solver_batch = 0

    state_size = property(lambda self: self._state_size)
    action_size = property(lambda self: self._action_size)
    error_state_size = property(lambda self: self._error_state_size)

    def _get_solver(self, B):
        if self._solver is None or self._solver_batch < B:
            prob = _problem_of(self.dynamics, self.cost)
            if prob.N != self.N:
                raise ValueError("reference trajectory has %d knots, controller horizon N = %d" % (prob.N + 1, self.N))
            self._solver = BatchedTrackingILQR(prob, B)
            self._solver_batch = B
        return self._solver

    def _options(self):
        return dict(line_search=False, rollout=self._rollout_mode, max_reg=self._mu_max)

    def _attach_al(self, solver, B):
        """A plain controller handed an ALConstrainedCost solves with that cost's current multipliers."""
        if not isinstance(self.cost, ALConstrainedCost):
            return False
        import torch
        f64 = dict(dtype=torch.float64, device=solver.device)
        lam = torch.as_tensor(np.broadcast_to(self.cost.lmbd[: self.N][None], (B, self.N, self.cost.constr_size)).copy(),
                              **f64)
        d = np.stack([np.diag(a) for a in self.cost.Imu[: self.N]])
        imu = torch.as_tensor(np.broadcast_to(d[None], (B,) + d.shape).copy(), **f64)
        solver.set_al(self.cost.constr.lb, self.cost.constr.ub, lam, imu)
        return True

    def fit_batch(self, x0s, us_init=None, n_iterations=100, tol_grad_norm=None, tol_d_norm=1e-6):
        """B independent fits on the GPU.  x0s: list of [q (4,4), xi (6,)] or (q [B,4,4], xi [B,6]);
        us_init [B,N,m] or [N,m] (shared) or None (zeros).  Returns a FitResult of device tensors."""
        if isinstance(x0s, tuple) and len(x0s) == 2 and np.ndim(x0s[0]) == 3:
            q, xi = np.asarray(x0s[0], float), np.asarray(x0s[1], float)
        else:
            q = np.stack([np.asarray(x[0], float) for x in x0s])
            xi = np.stack([np.asarray(x[1], float) for x in x0s])
        B = q.shape[0]
        if us_init is not None and np.ndim(us_init) == 2:
            us_init = np.broadcast_to(np.asarray(us_init, float), (B,) + np.shape(us_init)).copy()
        solver = self._get_solver(B)
        tol = self._default_tol if tol_grad_norm is None else tol_grad_norm
        al = self._attach_al(solver, B)
        try:
            return solver.fit_batch(q, xi, us_init, mode=self._mode, n_iterations=n_iterations, tol_grad_norm=tol,
                                    tol_d_norm=tol_d_norm, **self._options())
        finally:
            if al:
                solver.set_al(None)

    # one trajectory, iteration by iteration, so that the callback sees what the reference shows it
    def _fit_single(self, x0, us_init, n_iterations, tol_grad_norm, tol_d_norm, on_iteration, ms):
        q0, xi0 = _bridge.split_state(x0)
        solver = self._get_solver(1)
        us0 = np.asarray(us_init, float).reshape(1, self.N, self._action_size)
        J_hist, xs_hist, us_hist, grad_hist, defect_hist = [], [], [], [], []
        al = self._attach_al(solver, 1)
        res = solver.solve_begin(q0, xi0, us0, mode=self._mode, n_iterations=n_iterations, tol_grad_norm=tol_grad_norm,
                                 tol_d_norm=tol_d_norm, **self._options())
        solver.solve_peek()
        xs = _xs_list(_bridge.host(res.xs_q)[0], _bridge.host(res.xs_xi)[0])
        us = _bridge.host(res.us)[0].copy()
        xs_hist.append(list(xs))
        us_hist.append(us.copy())
        if ms:
            defect_hist.append(float(res.defect_hist[0, 0]))
        converged = False
        for it in range(int(n_iterations)):
            solver.solve_iterate(1)
            solver.solve_peek()
            iters, status, conv = int(res.iters[0]), int(res.status[0]), int(res.converged[0])
            grad = float(res.grad_hist[0, it])
            if not ms:
                grad_hist.append(grad)  # SS appends inside fit (traopt_controller.py:1938)
            if conv:  # gradient test fired: the reference breaks before the callback (:2528-2532, :1939-1942)
                converged = True
                if ms and self._append_grad_on_convergence:
                    grad_hist.append(grad)
                break
            if status == 1:
                warnings.warn(_MSG_MAXREG)
            accepted = status != 2
            xs = _xs_list(_bridge.host(res.xs_q)[0], _bridge.host(res.xs_xi)[0])
            us = _bridge.host(res.us)[0].copy()
            J_opt = float(res.J_hist[0, it])
            alpha = float(res.alpha_hist[0, it])
            mu = float(res.mu_hist[0, it])
            self._mu = mu
            if on_iteration:
                if ms:  # 15 positional arguments (:2621-2626)
                    on_iteration(it, xs, us, J_opt, accepted, converged, float(res.defect_hist[0, it + 1]), grad, alpha, mu,
                                 J_hist, xs_hist, us_hist, grad_hist, defect_hist)
                else:   # 12 positional arguments (:1996-2000)
                    on_iteration(it, xs, us, J_opt, accepted, converged, grad, alpha, mu, J_hist, xs_hist, us_hist)
            if not accepted:
                warnings.warn(_MSG_NODESCENT)
                break
            if iters <= it:  # non-finite cost: the device froze this trajectory
                break
        solver.solve_end()
        if al:
            solver.set_al(None)
        return xs, us, J_hist, xs_hist, us_hist, grad_hist, defect_hist


class iLQR_Tracking_SE3(_FusedController):
    """Finite Horizon Iterative Linear Quadratic Regulator for Exact SE3 Dynamics (single shooting,
    traopt_controller.py:1831-2349)."""
    _mode = "ss"
    _default_tol = 1e-3

    def __init__(self, dynamics, cost, N, max_reg=1e10, hessians=False, rollout='linear', debug=None):
        self._common_init(dynamics, cost, N, max_reg, hessians, rollout, debug)

    def fit(self, x0, us_init, n_iterations=100, tol_J=1e-6, tol_grad_norm=1e-3, on_iteration=None):
        xs, us, J_hist, xs_hist, us_hist, grad_hist, _ = self._fit_single(x0, us_init, n_iterations, tol_grad_norm, 0.0,
                                                                           on_iteration, ms=False)
        return xs, us, J_hist, xs_hist, us_hist, grad_hist


class iLQR_Tracking_SE3_MS(_FusedController):
    """Finite Horizon Multiple Shooting Iterative Linear Quadratic Regulator for Exact SE3 Dynamics
    (traopt_controller.py:2352-3136)."""
    _mode = "ms"
    _default_tol = 1e-6

    def __init__(self, dynamics, cost, N, q_ref, xi_ref, max_reg=1e10, hessians=False, line_search=False,
                 rollout='linear', debug=None):
        self._common_init(dynamics, cost, N, max_reg, hessians, rollout, debug)
        self._q_ref = q_ref
        self._xi_ref = xi_ref
        self._line_search = line_search
        self._defect_mu0 = 10.
        self._defect_rho = 0.5
        self._defect_gamma = 0.05
        self._defect_mu_min = self._defect_mu0
        self._defect_kappa = 1e-12

    xi_ref = property(lambda self: self._xi_ref)
    q_ref = property(lambda self: self._q_ref)

    def get_q_ref(self, i):
        return self._q_ref[i]

    def get_xi_ref(self, i):
        return self._xi_ref[i]

    def _options(self):
        return dict(line_search=self._line_search, rollout=self._rollout_mode, max_reg=self._mu_max)

    def fit(self, x0, us_init, n_iterations=100, tol_J=1e-6, tol_grad_norm=1e-6, tol_d_norm=1e-6, on_iteration=None):
        return self._fit_single(x0, us_init, n_iterations, tol_grad_norm, tol_d_norm, on_iteration, ms=True)


class AL_iLQR_Tracking_SE3_MS(BaseController):
    """Multiple shooting with input box constraints through an augmented Lagrangian
    (traopt_controller.py:3139-3293).  The reference class does not run at HEAD;
    this follows its source with the three breakages repaired: the inner fit returns 7 values, the
    inner callback has the MS 15-argument signature, q_ref/xi_ref are stored."""

    def __init__(self, dynamics, cost, constraints, N, q_ref, xi_ref, mu_scale=10., max_reg=1e10, hessians=False,
                 line_search=False, rollout='nonlinear', debug=None):
        self.dynamics = dynamics
        self.cost = cost
        self.constr = constraints
        self.N = N
        self._q_ref = q_ref
        self._xi_ref = xi_ref
        self._action_size = dynamics.action_size
        self._state_size = dynamics.state_size
        self._error_state_size = dynamics._error_state_size
        self._constr_size = constraints.constr_size
        self._mu0 = 1e-2
        self._mu_scale = mu_scale
        self._mu_max = 1e8
        self.al = ALConstrainedCost(cost, constraints, N)
        # the reference hard-codes rollout='nonlinear' for the inner solver (:3195)
        self.ilqr_solver = iLQR_Tracking_SE3_MS(dynamics, self.al, N, q_ref, xi_ref, max_reg=max_reg, hessians=hessians,
                                                line_search=line_search, rollout='nonlinear', debug=debug)

    xi_ref = property(lambda self: self._xi_ref)
    q_ref = property(lambda self: self._q_ref)
    state_size = property(lambda self: self._state_size)
    action_size = property(lambda self: self._action_size)

    def fit_batch(self, x0s, us_init=None, n_al_iters=100, n_ilqr_iters=200, tol_grad_norm=1e-6, tol_constr=1e-2,
                  on_outer=None):
        q = np.stack([np.asarray(x[0], float) for x in x0s])
        xi = np.stack([np.asarray(x[1], float) for x in x0s])
        B = q.shape[0]
        if us_init is not None and np.ndim(us_init) == 2:
            us_init = np.broadcast_to(np.asarray(us_init, float), (B,) + np.shape(us_init)).copy()
        solver = self.ilqr_solver._get_solver(B)
        return solver.al_fit_batch(q, xi, us_init, self.constr.lb, self.constr.ub, n_al_iters=n_al_iters,
                                   n_ilqr_iters=n_ilqr_iters, tol_grad_norm=tol_grad_norm, tol_constr=tol_constr,
                                   mu0=self._mu0, mu_scale=self._mu_scale, mu_max=self._mu_max,
                                   line_search=self.ilqr_solver._line_search, on_outer=on_outer)

    def fit(self, x0, us_init, n_al_iters=100, n_ilqr_iters=200, tol_J=1e-6, tol_grad_norm=1e-6, tol_constr=1e-2,
            on_iteration_al=None, on_iteration_ilqr=None):
        lmbd_hist, mu_hist, violation_hist, nactive_hist = [], [], [], []
        m = self._action_size

        def outer(iteration, res, lam, imu, mu):
            if not on_iteration_al:
                return
            us = _bridge.host(res.us)[0]
            g = np.concatenate([self.constr.lb - us, us - self.constr.ub], axis=1)
            constr_eval = np.vstack([g, np.zeros((1, 2 * m))])
            lam_h = np.vstack([_bridge.host(lam)[0], np.zeros((1, 2 * m))])
            imu_h = np.stack([np.diag(d) for d in np.vstack([_bridge.host(imu)[0], np.zeros((1, 2 * m))])])
            on_iteration_al(iteration, bool(np.max(constr_eval) < tol_constr), lam_h, imu_h, float(mu[0]), constr_eval,
                            lmbd_hist, mu_hist, violation_hist, nactive_hist)  # 10 positional arguments (:3253-3259)

        # the reference's inner fit ignores the outer tol_grad_norm and uses 1e-6 (:3238-3240)
        res, info = self.fit_batch([x0], np.asarray(us_init, float), n_al_iters, n_ilqr_iters, 1e-6, tol_constr,
                                   on_outer=outer)
        self.al.lmbd[: self.N] = _bridge.host(info["lmbd"])[0]
        self.al.mu = float(info["mu"][0])
        xs = _xs_list(_bridge.host(res.xs_q)[0], _bridge.host(res.xs_xi)[0])
        us = _bridge.host(res.us)[0]
        n = int(res.iters[0])
        J_hist = list(_bridge.host(res.J_hist)[0][:n])
        grad_hist = list(_bridge.host(res.grad_hist)[0][:n])
        return xs, us, J_hist, [], [], grad_hist, lmbd_hist, mu_hist, violation_hist, nactive_hist


# ---------------------------------------------------------------------------------------------------
# SO(3) controllers: same algorithm, states are [SO3, SO3Tangent] objects (traopt_controller.py:526-1824)
# ---------------------------------------------------------------------------------------------------
class _FusedControllerSO3(_FusedController):
    def _get_solver(self, B):
        if self._solver is None or self._solver_batch < B:
            if type(self.dynamics) not in (SO3Dynamics, Pendulum3dDyanmics) or \
                    type(self.cost) is not SO3TrackingQuadraticGaussNewtonCost:
                raise TypeError("the MI355X SO(3) path supports SO3Dynamics / Pendulum3dDyanmics with "
                                "SO3TrackingQuadraticGaussNewtonCost")
            prob = self.cost._embedded_problem(self.dynamics.J, self.dynamics.dt)
            if type(self.dynamics) is Pendulum3dDyanmics:
                prob.kind, prob.pend_mass, prob.pend_length = "pendulum3d", float(self.dynamics.m), float(self.dynamics.l)
            if prob.N != self.N:
                raise ValueError("reference trajectory has %d knots, controller horizon N = %d" % (prob.N + 1, self.N))
            self._solver = BatchedTrackingILQR(prob, B)
            self._solver_batch = B
        return self._solver

    def _attach_al(self, solver, B):
        return False

    @staticmethod
    def _wrap_xs(xs):
        from .traopt_utilis import SO3, SO3Tangent
        return [[SO3.from_matrix(q[:3, :3]), SO3Tangent(xi[:3])] for q, xi in xs]

    def _fit_so3(self, x0, us_init, n_iterations, tol_grad_norm, tol_d_norm, on_iteration, ms):
        from .traopt_dynamics import _so3_state
        q0, xi0 = _so3_state(x0)
        us6 = np.zeros((self.N, 6))
        us6[:, :3] = np.asarray(us_init, float).reshape(self.N, 3)
        cb = None
        if on_iteration:
            def cb(it, xs, us, *rest):
                rest = list(rest)
                on_iteration(it, self._wrap_xs(xs), us[:, :3].copy(), *rest)
        self._action_size_embedded = 6
        saved = self._action_size
        self._action_size = 6
        try:
            xs, us, J_hist, xs_hist, us_hist, grad_hist, defect_hist = self._fit_single(
                [q0[0], xi0[0]], us6, n_iterations, tol_grad_norm, tol_d_norm, cb, ms)
        finally:
            self._action_size = saved
        return self._wrap_xs(xs), us[:, :3].copy(), J_hist, xs_hist, us_hist, grad_hist, defect_hist

    def fit_batch(self, x0s, us_init=None, n_iterations=100, tol_grad_norm=None, tol_d_norm=1e-6):
        """B independent SO(3) fits; returns the FitResult in the embedded SE(3) layout
        (rotation = xs_q[..., :3, :3], body rate = xs_xi[..., :3], torque = us[..., :3])."""
        from .traopt_dynamics import _so3_state
        st = [_so3_state(x) for x in x0s]
        q = np.concatenate([a for a, _ in st]); xi = np.concatenate([b for _, b in st])
        B = q.shape[0]
        us6 = None
        if us_init is not None:
            u = np.asarray(us_init, float)
            u = np.broadcast_to(u, (B,) + u.shape[-2:]) if u.ndim == 2 else u
            us6 = np.zeros((B, self.N, 6)); us6[:, :, :3] = u
        solver = self._get_solver(B)
        tol = self._default_tol if tol_grad_norm is None else tol_grad_norm
        return solver.fit_batch(q, xi, us6, mode=self._mode, n_iterations=n_iterations, tol_grad_norm=tol,
                                tol_d_norm=tol_d_norm, **self._options())


class iLQR_Tracking_SO3(_FusedControllerSO3):
    """Single-shooting iLQR on SO(3) (traopt_controller.py:526-1026)."""
    _mode = "ss"
    _default_tol = 1e-6

    def __init__(self, dynamics, cost, N, max_reg=1e10, hessians=False, rollout='nonlinear', debug=None):
        self._common_init(dynamics, cost, N, max_reg, hessians, rollout, debug)

    def fit(self, x0, us_init, n_iterations=100, tol_J=1e-6, tol_grad_norm=1e-6, on_iteration=None):
        xs, us, J_hist, xs_hist, us_hist, grad_hist, _ = self._fit_so3(x0, us_init, n_iterations, tol_grad_norm, 0.0,
                                                                        on_iteration, ms=False)
        return xs, us, J_hist, xs_hist, us_hist, grad_hist


class iLQR_Tracking_SO3_MS(_FusedControllerSO3):
    """Multiple-shooting iLQR on SO(3) (traopt_controller.py:1029-1824): 13 line-search alphas."""
    _mode = "ms"
    _default_tol = 1e-6
    _append_grad_on_convergence = True

    def __init__(self, dynamics, cost, N, q_ref, xi_ref, max_reg=1e10, hessians=False, line_search=False,
                 rollout='linear', debug=None):
        self._common_init(dynamics, cost, N, max_reg, hessians, rollout, debug)
        self._q_ref = q_ref
        self._xi_ref = xi_ref
        self._line_search = line_search
        self._defect_mu0 = 10.
        self._defect_rho = 0.5
        self._defect_gamma = 0.05
        self._defect_mu_min = self._defect_mu0
        self._defect_kappa = 1e-14

    xi_ref = property(lambda self: self._xi_ref)
    q_ref = property(lambda self: self._q_ref)

    def _options(self):
        return dict(line_search=self._line_search, rollout=self._rollout_mode, max_reg=self._mu_max)

    def fit(self, x0, us_init, n_iterations=100, tol_J=1e-6, tol_grad_norm=1e-6, tol_d_norm=1e-6, on_iteration=None):
        out = self._fit_so3(x0, us_init, n_iterations, tol_grad_norm, tol_d_norm, on_iteration, ms=True)
        return out


# ---------------------------------------------------------------------------------------------------
# Euclidean-space controller (traopt_controller.py:42-520): config 1 of BASELINE.json, host plumbing
# ---------------------------------------------------------------------------------------------------
class PDViolationError(Exception):
    """Custom exception class for handling positive definite violation errors (traopt_controller.py:35-37)."""


class iLQR(BaseController):
    """Finite Horizon Iterative Linear Quadratic Regulator on a Euclidean state (traopt_controller.py:42-520),
    with the true-DDP tensor terms when hessians=True (:487-490).

    Generic plugin path: any BaseDynamics / BaseCost works through the per-knot methods; AutoDiffDynamics /
    AutoDiffCost additionally expose knot-batched derivatives, which _forward_rollout uses to evaluate the
    N Jacobians of a rollout in one torch.func.vmap call instead of N Python calls.  The solver arithmetic
    (NumPy, fp64) follows the reference line by line, including its line-search and regularisation rules."""

    def __init__(self, dynamics, cost, N, max_reg=1e10, hessians=False):
        self.dynamics = dynamics
        self.cost = cost
        self.N = N
        self._use_hessians = hessians and dynamics.has_hessians
        if hessians and not dynamics.has_hessians:
            warnings.warn("hessians requested but are unavailable in dynamics")
        self._mu = 1.0
        self._mu_min = 1e-6
        self._mu_max = max_reg
        self._delta_0 = 2.0
        self._delta = self._delta_0
        self._action_size = dynamics.action_size
        self._state_size = dynamics.state_size
        self._k = np.zeros((N, self._action_size))
        self._K = np.zeros((N, self._action_size, self._state_size))

    def fit(self, x0, us_init, n_iterations=100, tol_J=1e-6, tol_grad_norm=1e-3, on_iteration=None):
        self._mu = 1.0
        self._delta = self._delta_0
        alphas = 1.1 ** (-np.arange(10) ** 2)
        us = np.array(us_init, dtype=float)
        k, K = self._k, self._K
        J_hist, xs_hist, us_hist = [], [], []
        changed = True
        converged = False
        alpha = alphas[0]
        grad_wrt_input_norm = np.inf
        for iteration in range(n_iterations):
            accepted = False
            if changed:
                (xs, F_x, F_u, L, L_x, L_u, L_xx, L_ux, L_uu, F_xx, F_ux, F_uu) = self._forward_rollout(x0, us)
                J_opt = L.sum()
                changed = False
            try:
                k, K = self._backward_pass(F_x, F_u, L_x, L_u, L_xx, L_ux, L_uu, F_xx, F_ux, F_uu)
                _, grad_wrt_input_norm = self._gradient_wrt_control(F_x, F_u, L_x, L_u)  # alpha independent (:166)
                for alpha in alphas:
                    xs_new, us_new = self._control(xs, us, k, K, alpha)
                    J_new = self._trajectory_cost(xs_new, us_new)
                    if grad_wrt_input_norm < tol_grad_norm:
                        converged = True
                        accepted = True
                        break
                    if J_new < J_opt:
                        if np.abs((J_opt - J_new) / J_opt) < tol_J:
                            converged = True
                        J_opt = J_new
                        xs = xs_new
                        us = us_new
                        changed = True
                        self._delta = min(1.0, self._delta) / self._delta_0
                        self._mu *= self._delta
                        if self._mu <= self._mu_min:
                            self._mu = 0.0
                        accepted = True
                        break
            except np.linalg.LinAlgError as e:
                warnings.warn(str(e))
            if not accepted:
                self._delta = max(1.0, self._delta) * self._delta_0
                self._mu = max(self._mu_min, self._mu * self._delta)
                if self._mu_max and self._mu >= self._mu_max:
                    warnings.warn("exceeded max regularization term")
                    break
            if on_iteration:
                on_iteration(iteration, xs, us, J_opt, accepted, converged, grad_wrt_input_norm, alpha, self._mu,
                             J_hist, xs_hist, us_hist)
            if converged:
                break
        self._k = k
        self._K = K
        self._nominal_xs = xs
        self._nominal_us = us
        return xs, us, J_hist, xs_hist, us_hist

    def _control(self, xs, us, k, K, alpha=1.0):
        xs_new = np.zeros_like(xs)
        us_new = np.zeros_like(us)
        xs_new[0] = xs[0].copy()
        for i in range(self.N):
            us_new[i] = us[i] + alpha * k[i] + K[i].dot(xs_new[i] - xs[i])
            xs_new[i + 1] = self.dynamics.f(xs_new[i], us_new[i], i)
        return xs_new, us_new

    def _trajectory_cost(self, xs, us):
        if hasattr(self.cost, "batch"):
            J = float(np.sum(self.cost.batch("l", xs[:-1], us)))
        else:
            J = sum(self.cost.l(x, u, i) for i, (x, u) in enumerate(zip(xs[:-1], us)))
        return J + self.cost.l(xs[-1], None, self.N, terminal=True)

    def _forward_rollout(self, x0, us):
        n, m, N = self.dynamics.state_size, self.dynamics.action_size, us.shape[0]
        xs = np.empty((N + 1, n))
        xs[0] = np.asarray(x0, dtype=float)
        for i in range(N):
            xs[i + 1] = self.dynamics.f(xs[i], us[i], i)
        dyn_b, cost_b = hasattr(self.dynamics, "batch"), hasattr(self.cost, "batch")

        def dyn(which, shape):
            if dyn_b:
                return self.dynamics.batch(which, xs[:-1], us).reshape((N,) + shape)
            return np.stack([np.asarray(getattr(self.dynamics, which)(xs[i], us[i], i)).reshape(shape) for i in range(N)])

        def cst(which, shape):
            if cost_b:
                return self.cost.batch(which, xs[:-1], us).reshape((N,) + shape)
            return np.stack([np.asarray(getattr(self.cost, which)(xs[i], us[i], i, terminal=False)).reshape(shape)
                             for i in range(N)])

        F_x, F_u = dyn("f_x", (n, n)), dyn("f_u", (n, m))
        F_xx = F_ux = F_uu = None
        if self._use_hessians:
            F_xx, F_ux, F_uu = dyn("f_xx", (n, n, n)), dyn("f_ux", (n, m, n)), dyn("f_uu", (n, m, m))
        L = np.empty(N + 1)
        L_x = np.empty((N + 1, n))
        L_xx = np.empty((N + 1, n, n))
        L[:N] = cst("l", ())
        L_x[:N] = cst("l_x", (n,))
        L_u = cst("l_u", (m,))
        L_xx[:N] = cst("l_xx", (n, n))
        L_ux = cst("l_ux", (m, n))
        L_uu = cst("l_uu", (m, m))
        x = xs[-1]
        L[-1] = self.cost.l(x, None, N, terminal=True)
        L_x[-1] = self.cost.l_x(x, None, N, terminal=True)
        L_xx[-1] = self.cost.l_xx(x, None, N, terminal=True)
        return xs, F_x, F_u, L, L_x, L_u, L_xx, L_ux, L_uu, F_xx, F_ux, F_uu

    def _backward_pass(self, F_x, F_u, L_x, L_u, L_xx, L_ux, L_uu, F_xx=None, F_ux=None, F_uu=None):
        V_x = L_x[-1]
        V_xx = L_xx[-1]
        k = np.empty_like(self._k)
        K = np.empty_like(self._K)
        for i in range(self.N - 1, -1, -1):
            if self._use_hessians:
                Q_x, Q_u, Q_xx, Q_ux, Q_uu = self._Q(F_x[i], F_u[i], L_x[i], L_u[i], L_xx[i], L_ux[i], L_uu[i], V_x,
                                                     V_xx, F_xx[i], F_ux[i], F_uu[i])
            else:
                Q_x, Q_u, Q_xx, Q_ux, Q_uu = self._Q(F_x[i], F_u[i], L_x[i], L_u[i], L_xx[i], L_ux[i], L_uu[i], V_x,
                                                     V_xx)
            k[i] = -np.linalg.solve(Q_uu, Q_u)   # the reference tests is_pos_def here and carries on (:398-399)
            K[i] = -np.linalg.solve(Q_uu, Q_ux)
            V_x = Q_x + K[i].T.dot(Q_uu).dot(k[i])
            V_x += K[i].T.dot(Q_u) + Q_ux.T.dot(k[i])
            V_xx = Q_xx + K[i].T.dot(Q_uu).dot(K[i])
            V_xx += K[i].T.dot(Q_ux) + Q_ux.T.dot(K[i])
            V_xx = 0.5 * (V_xx + V_xx.T)
        return np.array(k), np.array(K)

    def _Q(self, f_x, f_u, l_x, l_u, l_xx, l_ux, l_uu, V_x, V_xx, f_xx=None, f_ux=None, f_uu=None):
        Q_x = l_x + f_x.T.dot(V_x)
        Q_u = l_u + f_u.T.dot(V_x)
        Q_xx = l_xx + f_x.T.dot(V_xx).dot(f_x)
        reg = self._mu * np.eye(self.dynamics.state_size)
        Q_ux = l_ux + f_u.T.dot(V_xx + reg).dot(f_x)
        Q_uu = l_uu + f_u.T.dot(V_xx + reg).dot(f_u)
        if self._use_hessians:
            Q_xx = Q_xx + np.tensordot(V_x, f_xx, axes=1)
            Q_ux = Q_ux + np.tensordot(V_x, f_ux, axes=1)
            Q_uu = Q_uu + np.tensordot(V_x, f_uu, axes=1)
        return Q_x, Q_u, Q_xx, Q_ux, Q_uu

    def _gradient_wrt_control(self, F_x, F_u, L_x, L_u):
        g = np.zeros((self.N, self._action_size))
        p = L_x[self.N]
        g_norm_sum = 0
        for t in range(self.N - 1, -1, -1):
            g[t] = L_u[t] + np.matmul(F_u[t].T, p)
            p = L_x[t] + np.matmul(F_x[t].T, p)
            g_norm_sum = g_norm_sum + np.linalg.norm(g[t])
        return g, g_norm_sum / self.N
