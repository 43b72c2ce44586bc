"""Controllers with the reference's interface (traoptlibrary/traopt_controller.py).

BaseController :14-40, iLQR_Tracking_SE3 :1831-2349, iLQR_Tracking_SE3_MS :2352-3136,
AL_iLQR_Tracking_SE3_MS :3139-3293.  ``fit`` keeps the reference's signature, return tuple and
callback protocol for one trajectory; ``fit_batch`` (new) solves B initial states at once.  For the
closed-form plugin classes both dispatch the whole iteration loop to the HIP extension (no per-knot
Python loop, no CPU twin of the kernels).  Any other plugin -- a subclass, an overridden method, a
hand-written BaseDynamics / BaseCost -- takes the generic per-knot callback loop of _generic_lie.py,
as SURVEY.md §8b prescribes."""
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


def _fusable(dynamics, cost):
    """True when (dynamics, cost) is one of the closed-form triples the HIP path implements: exact types only,
    a subclass may override any method (SURVEY.md §8b 'What calls it')."""
    base = cost.cost if isinstance(cost, ALConstrainedCost) else cost
    return type(dynamics) in _KIND and type(base) is SE3TrackingQuadraticGaussNewtonCost


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

    def _is_fusable(self):
        return _fusable(self.dynamics, self.cost)

    def _fit_generic(self, x0, us_init, n_iterations, tol_grad_norm, tol_d_norm, on_iteration, ms):
        """Plugins outside the closed-form set: per-knot callback loop on the host (_generic_lie.py)."""
        from ._generic_lie import GenericLieILQR
        opt = self._options()
        g = GenericLieILQR(self.dynamics, self.cost, self.N, "ms" if ms else "ss", max_reg=opt["max_reg"],
                           line_search=opt["line_search"], rollout=opt["rollout"])
        out = g.fit(x0, us_init, getattr(self, "_q_ref", None), getattr(self, "_xi_ref", None), n_iterations,
                    tol_grad_norm, tol_d_norm, on_iteration, self._append_grad_on_convergence)
        self._mu = g.mu
        return out

    # one trajectory, iteration by iteration, so that the callback sees what the reference shows it
    def _fit_single(self, x0, us_init, n_iterations, tol_grad_norm, tol_d_norm, on_iteration, ms):
        if not self._is_fusable():
            return self._fit_generic(x0, us_init, n_iterations, tol_grad_norm, tol_d_norm, on_iteration, ms)
        q0, xi0 = _bridge.split_state(x0)
        solver = self._get_solver(1)
        us0 = np.asarray(us_init, float).reshape(1, self.N, self._action_size)
        J_hist, xs_hist, us_hist, grad_hist, defect_hist = [], [], [], [], []
        al = self._attach_al(solver, 1)
        begun = False
        try:
            res = solver.solve_begin(q0, xi0, us0, mode=self._mode, n_iterations=n_iterations,
                                     tol_grad_norm=tol_grad_norm, tol_d_norm=tol_d_norm, **self._options())
            begun = True
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
        finally:  # a raising callback must not leave the handle mid-solve or the multipliers attached
            if begun:
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
    def _is_fusable(self):
        return True  # the SO(3) controllers have no generic path: _get_solver rejects other plugin types

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


class _Expansion:
    """Second-order model of one rollout in stacked, homogeneous form.

    z = (x, u, 1): `G[i]` ((n+1) x (n+m+1)) maps z_i to (x_{i+1}, 1) to first order, `H[i]` ((n+m+1)^2) is the
    Hessian of the stage cost in z with the gradient in its last row / column, `T[i]` (n x (n+m) x (n+m), DDP
    only) the second derivative of the dynamics.  One backward sweep over these carries the value function
    [[V_xx, V_x], [V_x^T, .]] and the adjoint of the control gradient together."""

    __slots__ = ("xs", "J", "G", "H", "T", "HN", "fu", "n", "m")


class iLQR(BaseController):
    """Finite Horizon Iterative Linear Quadratic Regulator on a Euclidean state, with the true-DDP tensor terms
    when hessians=True.  Same interface, acceptance rule, regularisation schedule and callback protocol as the
    reference class (traopt_controller.py:42-520: `fit` :83-222, the 10 step sizes :118, the acceptance /
    regularisation bookkeeping :160-207, the 12-argument callback :209-211); the arithmetic is organised
    differently -- BASELINE config 1 is host plumbing here, not the hot path:

    * derivatives of a whole rollout come from the knot-batched plugin methods (`AutoDiff*.batch`: one
      torch.func.vmap call per quantity) when the plugin has them, per-knot calls otherwise;
    * the backward pass is ONE sweep in homogeneous coordinates: Q = H_i + G_i^T V G_i (+ mu-regularised input
      rows, + V_x . T_i for DDP) is a single congruence per knot, the gain block [K | k] one solve, the new value
      function one Schur-type update -- and the adjoint p_t of the gradient test rides along as an extra vector,
      the way the device kernel K2 carries it in a spare lane."""

    _ALPHAS = 1.1 ** (-np.arange(10) ** 2)

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

    # ---- plugin access: knot-batched when available ------------------------------------------------
    def _stack(self, plugin, which, xs, us, shape, **kw):
        N = us.shape[0]
        if hasattr(plugin, "batch"):
            return np.asarray(plugin.batch(which, xs[:N], us)).reshape((N,) + shape)
        fn = getattr(plugin, which)
        return np.stack([np.asarray(fn(xs[i], us[i], i, **kw)).reshape(shape) for i in range(N)])

    def _simulate(self, x0, us):
        xs = np.empty((us.shape[0] + 1, self._state_size))
        xs[0] = np.asarray(x0, dtype=float)
        for i, u in enumerate(us):
            xs[i + 1] = self.dynamics.f(xs[i], u, i)
        return xs

    def _trajectory_cost(self, xs, us):
        stage = self._stack(self.cost, "l", xs, us, (), terminal=False)
        return float(stage.sum()) + float(self.cost.l(xs[-1], None, self.N, terminal=True))

    def _expand(self, xs, us):
        n, m, N = self._state_size, self._action_size, self.N
        dyn, cost = self.dynamics, self.cost
        e = _Expansion()
        e.xs, e.n, e.m = xs, n, m
        fx = self._stack(dyn, "f_x", xs, us, (n, n))
        e.fu = self._stack(dyn, "f_u", xs, us, (n, m))
        e.G = np.zeros((N, n + 1, n + m + 1))
        e.G[:, :n, :n] = fx
        e.G[:, :n, n:n + m] = e.fu
        e.G[:, n, n + m] = 1.0
        kw = dict(terminal=False)
        e.H = np.zeros((N, n + m + 1, n + m + 1))
        e.H[:, :n, :n] = self._stack(cost, "l_xx", xs, us, (n, n), **kw)
        lux = self._stack(cost, "l_ux", xs, us, (m, n), **kw)
        e.H[:, n:n + m, :n] = lux
        e.H[:, :n, n:n + m] = np.swapaxes(lux, 1, 2)
        e.H[:, n:n + m, n:n + m] = self._stack(cost, "l_uu", xs, us, (m, m), **kw)
        g = np.concatenate([self._stack(cost, "l_x", xs, us, (n,), **kw), self._stack(cost, "l_u", xs, us, (m,), **kw)], axis=1)
        e.H[:, :n + m, n + m] = g
        e.H[:, n + m, :n + m] = g
        stage = self._stack(cost, "l", xs, us, (), **kw)
        xN = xs[-1]
        e.J = float(stage.sum()) + float(cost.l(xN, None, N, terminal=True))
        e.HN = np.zeros((n + 1, n + 1))
        e.HN[:n, :n] = cost.l_xx(xN, None, N, terminal=True)
        e.HN[:n, n] = e.HN[n, :n] = cost.l_x(xN, None, N, terminal=True)
        e.T = None
        if self._use_hessians:
            e.T = np.zeros((N, n, n + m, n + m))
            fux = self._stack(dyn, "f_ux", xs, us, (n, m, n))
            e.T[:, :, :n, :n] = self._stack(dyn, "f_xx", xs, us, (n, n, n))
            e.T[:, :, n:, :n] = fux
            e.T[:, :, :n, n:] = np.swapaxes(fux, 2, 3)
            e.T[:, :, n:, n:] = self._stack(dyn, "f_uu", xs, us, (n, m, m))
        return e

    def _sweep(self, e, mu):
        """Gains [K | k] of every knot and the mean norm of the control gradient dJ/du_t."""
        n, m, N = e.n, e.m, self.N
        V = e.HN.copy()                      # [[V_xx, V_x], [V_x^T, 0]]
        p = e.HN[:n, n].copy()               # adjoint of the open-loop gradient
        gains = np.empty((N, m, n + 1))
        gsum = 0.0
        for i in range(N - 1, -1, -1):
            G, fu = e.G[i], e.fu[i]
            Q = e.H[i] + G.T @ V @ G
            if mu:                           # regularisation acts on the input rows only: F_u^T (V_xx + mu I) [F_x F_u]
                Q[n:n + m, :n + m] += mu * (fu.T @ G[:n, :n + m])
                Q[:n, n:n + m] = Q[n:n + m, :n].T
            if e.T is not None:
                Q[:n + m, :n + m] += np.tensordot(V[:n, n], e.T[i], axes=1)
            gu = e.H[i][n:n + m, n + m] + fu.T @ p
            gsum += float(np.linalg.norm(gu))
            p = e.H[i][:n, n + m] + G[:n, :n].T @ p
            Quu = Q[n:n + m, n:n + m]
            Qua = np.concatenate([Q[n:n + m, :n], Q[n:n + m, n + m:]], axis=1)      # [Q_ux | Q_u]
            Kk = -np.linalg.solve(Quu, Qua)
            gains[i] = Kk
            Qaa = np.empty((n + 1, n + 1))
            Qaa[:n, :n] = Q[:n, :n]
            Qaa[:n, n] = Qaa[n, :n] = Q[:n, n + m]
            Qaa[n, n] = Q[n + m, n + m]
            cross = Kk.T @ Qua
            V = Qaa + Kk.T @ Quu @ Kk + cross + cross.T
            V[:n, :n] = 0.5 * (V[:n, :n] + V[:n, :n].T)
        return gains[:, :, n].copy(), gains[:, :, :n].copy(), gsum / N

    def _closed_loop(self, xs, us, k, K, alpha):
        xs_new, us_new = np.empty_like(xs), np.empty_like(us)
        xs_new[0] = xs[0]
        for i in range(self.N):
            us_new[i] = us[i] + alpha * k[i] + K[i] @ (xs_new[i] - xs[i])
            xs_new[i + 1] = self.dynamics.f(xs_new[i], us_new[i], i)
        return xs_new, us_new

    def _relax(self):
        self._delta = min(1.0, self._delta) / self._delta_0
        self._mu *= self._delta
        if self._mu <= self._mu_min:
            self._mu = 0.0

    def _tighten(self):
        self._delta = max(1.0, self._delta) * self._delta_0
        self._mu = max(self._mu_min, self._mu * self._delta)
        return bool(self._mu_max and self._mu >= self._mu_max)

    def fit(self, x0, us_init, n_iterations=100, tol_J=1e-6, tol_grad_norm=1e-3, on_iteration=None):
        self._mu, self._delta = 1.0, self._delta_0
        us = np.array(us_init, dtype=float)
        xs = self._simulate(x0, us)
        J_hist, xs_hist, us_hist = [], [], []
        k, K = self._k, self._K
        model, grad, alpha, J_opt = None, np.inf, self._ALPHAS[0], None
        for iteration in range(n_iterations):
            if model is None:                # the trajectory moved: new expansion
                model = self._expand(xs, us)
                J_opt = model.J
            accepted = converged = False
            try:
                k, K, grad = self._sweep(model, self._mu)
                for alpha in self._ALPHAS:
                    xs_try, us_try = self._closed_loop(xs, us, k, K, alpha)
                    J_try = self._trajectory_cost(xs_try, us_try)
                    if grad < tol_grad_norm:             # stationary: stop without moving
                        accepted = converged = True
                        break
                    if J_try < J_opt:
                        converged = abs((J_opt - J_try) / J_opt) < tol_J
                        xs, us, J_opt, model = xs_try, us_try, J_try, None
                        self._relax()
                        accepted = True
                        break
            except np.linalg.LinAlgError as err:
                warnings.warn(str(err))
            if not accepted and self._tighten():
                warnings.warn("exceeded max regularization term")
                break
            if on_iteration:
                on_iteration(iteration, xs, us, J_opt, accepted, converged, grad, alpha, self._mu, J_hist, xs_hist,
                             us_hist)
            if converged:
                break
        self._k, self._K = k, K
        self._nominal_xs, self._nominal_us = xs, us
        return xs, us, J_hist, xs_hist, us_hist
