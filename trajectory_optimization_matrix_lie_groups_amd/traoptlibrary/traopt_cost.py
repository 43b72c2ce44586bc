"""Cost plugins with the reference's interface (traoptlibrary/traopt_cost.py).

BaseCost :14-110, SE3TrackingQuadraticGaussNewtonCost :570-867 (and the names older scripts import),
ALConstrainedCost :1173-1320."""
import abc

import numpy as np

from . import _bridge


class BaseCost():
    """Instantaneous Cost (traopt_cost.py:14-110)."""

    @abc.abstractmethod
    def l(self, x, u, i, terminal=False):
        raise NotImplementedError

    @abc.abstractmethod
    def l_x(self, x, u, i, terminal=False):
        raise NotImplementedError

    @abc.abstractmethod
    def l_u(self, x, u, i, terminal=False):
        raise NotImplementedError

    @abc.abstractmethod
    def l_xx(self, x, u, i, terminal=False):
        raise NotImplementedError

    @abc.abstractmethod
    def l_ux(self, x, u, i, terminal=False):
        raise NotImplementedError

    @abc.abstractmethod
    def l_uu(self, x, u, i, terminal=False):
        raise NotImplementedError


class SE3TrackingQuadraticGaussNewtonCost(BaseCost):
    """||Log(X Xref^-1)||^2_Q1 + ||xi - xi_ref||^2_Q2 + ||u||^2_R, terminal with P
    (traopt_cost.py:570-867)."""

    def __init__(self, Q, R, P, q_ref, xi_ref, state_size=(6, 6), action_size=6, **kwargs):
        self._state_size = state_size[0] + state_size[1]
        self._error_state_size = state_size[0]
        self._vel_state_size = state_size[1]
        self._action_size = action_size
        self._q_ref_mats = np.asarray(q_ref, dtype=float)
        self._xi_ref = np.asarray(xi_ref, dtype=float)
        self._Q = np.asarray(Q, dtype=float)
        self._R = np.asarray(R, dtype=float)
        self._P = np.asarray(P, dtype=float)
        self._probe_solver = None
        self._al = None

    state_size = property(lambda self: self._state_size)
    error_state_size = property(lambda self: self._error_state_size)
    vel_state_size = property(lambda self: self._vel_state_size)
    action_size = property(lambda self: self._action_size)
    Q = property(lambda self: self._Q)
    R = property(lambda self: self._R)
    P = property(lambda self: self._P)

    @property
    def N(self):
        return self._q_ref_mats.shape[0] - 1

    def _probe(self):
        if self._probe_solver is None:
            self._probe_solver = _bridge.cost_probe(self._Q, self._R, self._P, self._q_ref_mats, self._xi_ref,
                                                    self._action_size)
        return self._probe_solver

    def _eval(self, x, u, i, terminal=False):
        if terminal and int(i) != self.N:
            raise ValueError("terminal cost is defined at the last knot (i = N)")
        q, xi = _bridge.split_state(x)
        return self._probe().eval_knot(self.N if terminal else int(i), q, xi, None if u is None else
                                       np.asarray(u, float).reshape(1, self._action_size))

    def _err(self, x, i):
        """(Log(X Xref_i^-1) [w, v], xi - xi_ref_i) (traopt_cost.py:659-673)."""
        e = _bridge.host(self._eval(x, np.zeros(self._action_size), i, terminal=(int(i) == self.N))["err"])[0]
        return e[:6], e[6:]

    def l(self, x, u, i, terminal=False):
        return float(_bridge.host(self._eval(x, u, i, terminal)["l"])[0])

    def l_x(self, x, u, i, terminal=False):
        return _bridge.host(self._eval(x, u, i, terminal)["lx"])[0]

    def l_u(self, x, u, i, terminal=False):
        if terminal:
            return np.zeros(self._action_size)
        return _bridge.host(self._eval(x, u, i)["lu"])[0]

    def l_xx(self, x, u, i, terminal=False):
        return _bridge.host(self._eval(x, u, i, terminal)["lxx"])[0]

    def l_ux(self, x, u, i, terminal=False):
        return np.zeros((self.action_size, self.state_size))  # traopt_cost.py:853

    def l_uu(self, x, u, i, terminal=False):
        if terminal:
            return 2 * self._R
        return _bridge.host(self._eval(x, u, i)["luu"])[0]


# names older scripts of the reference import
ErrorStateSE3TrackingQuadraticGaussNewtonCost = SE3TrackingQuadraticGaussNewtonCost
DroneTrackingQuadraticGaussNewtonCost = SE3TrackingQuadraticGaussNewtonCost  # the original cannot be constructed (App. C-Q8)


class ALConstrainedCost(BaseCost):
    """LA = l + lambda^T g + g^T I_mu g / 2 (traopt_cost.py:1173-1320) for an InputConstraint."""

    def __init__(self, cost, constraints, N, state_size=(6, 6), action_size=6, **kwargs):
        self._state_size = state_size[0] + state_size[1]
        self._error_state_size = state_size[0]
        self._vel_state_size = state_size[1]
        self._action_size = cost.action_size
        self._constr_size = constraints.constr_size
        self.constr = constraints
        self.cost = cost
        self.N = N
        self.lmbd = np.zeros((N + 1, self._constr_size))
        self.mu = 0.
        self.Imu = np.zeros((N + 1, self._constr_size, self._constr_size))

    state_size = property(lambda self: self._state_size)
    error_state_size = property(lambda self: self._error_state_size)
    vel_state_size = property(lambda self: self._vel_state_size)
    action_size = property(lambda self: self._action_size)
    constr_size = property(lambda self: self._constr_size)

    def _eval(self, x, u, i, terminal=False):
        import torch
        solver = self.cost._probe()
        if terminal:
            return self.cost._eval(x, u, i, terminal=True)
        f64 = dict(dtype=torch.float64, device=solver.device)
        lam = torch.as_tensor(self.lmbd[: self.N][None], **f64).contiguous()
        imu = torch.as_tensor(np.stack([np.diag(a) for a in self.Imu[: self.N]])[None], **f64).contiguous()
        solver.set_al(self.constr.lb, self.constr.ub, lam, imu)
        try:
            return self.cost._eval(x, u, i)
        finally:
            solver.set_al(None)

    def l(self, x, u, i, terminal=False):
        return float(_bridge.host(self._eval(x, u, i, terminal)["l"])[0])

    def l_x(self, x, u, i, terminal=False):
        return _bridge.host(self._eval(x, u, i, terminal)["lx"])[0]  # g_x = 0

    def l_u(self, x, u, i, terminal=False):
        if terminal:
            return np.zeros(self._action_size)
        return _bridge.host(self._eval(x, u, i)["lu"])[0]

    def l_xx(self, x, u, i, terminal=False):
        return _bridge.host(self._eval(x, u, i, terminal)["lxx"])[0]

    def l_ux(self, x, u, i, terminal=False):
        return np.zeros((self.action_size, self.state_size))

    def l_uu(self, x, u, i, terminal=False):
        if terminal:
            return 2 * self.cost.R
        return _bridge.host(self._eval(x, u, i)["luu"])[0]


class SO3TrackingQuadraticGaussNewtonCost(BaseCost):
    """||Log(R Rref^T)||^2_Q1 + ||w - w_ref||^2_Q2 + ||u||^2_R on SO(3) (traopt_cost.py:280-564), with its
    terminal-weight quirk kept: l and l_x always use Q, only l_xx switches to P."""

    def __init__(self, Q, R, P, q_ref, xi_ref, state_size=(3, 3), action_size=3, **kwargs):
        self._state_size = state_size[0] + state_size[1]
        self._pos_state_size = state_size[0]
        self._vel_state_size = state_size[1]
        self._action_size = action_size
        self._q_ref_mats = np.asarray(q_ref, dtype=float)   # (N+1, 3, 3)
        self._xi_ref = np.asarray(xi_ref, dtype=float)      # (N+1, 3)
        self._Q = np.asarray(Q, dtype=float)
        self._R = np.asarray(R, dtype=float)
        self._P = np.asarray(P, dtype=float)
        self._probe_solver = None

    state_size = property(lambda self: self._state_size)
    pos_state_size = property(lambda self: self._pos_state_size)
    vel_state_size = property(lambda self: self._vel_state_size)
    action_size = property(lambda self: self._action_size)
    Q = property(lambda self: self._Q)
    R = property(lambda self: self._R)
    P = property(lambda self: self._P)

    @property
    def N(self):
        return self._q_ref_mats.shape[0] - 1

    def _embedded_problem(self, J3=None, dt=1.0):
        from ..solver import embed_so3
        return embed_so3(np.eye(3) if J3 is None else J3, dt, self._Q, self._R, self._P, self._q_ref_mats, self._xi_ref)

    def _probe(self):
        if self._probe_solver is None:
            from ..solver import BatchedTrackingILQR
            self._probe_solver = BatchedTrackingILQR(self._embedded_problem(), 1)
        return self._probe_solver

    def _eval(self, x, u, i, terminal=False):
        from .traopt_dynamics import _so3_state
        if terminal and int(i) != self.N:
            raise ValueError("terminal cost is defined at the last knot (i = N)")
        q, xi = _so3_state(x)
        u6 = None if u is None else np.r_[np.asarray(u, float).reshape(3), 0, 0, 0].reshape(1, 6)
        return self._probe().eval_knot(self.N if terminal else int(i), q, xi, u6)

    _IDX = [0, 1, 2, 6, 7, 8]

    def _err(self, x, i):
        e = _bridge.host(self._eval(x, np.zeros(3), i, terminal=(int(i) == self.N))["err"])[0]
        return e[:3], e[6:9]

    def l(self, x, u, i, terminal=False):
        return float(_bridge.host(self._eval(x, u, i, terminal)["l"])[0])

    def l_x(self, x, u, i, terminal=False):
        return _bridge.host(self._eval(x, u, i, terminal)["lx"])[0][self._IDX]

    def l_u(self, x, u, i, terminal=False):
        return 2 * self._R @ np.asarray(u, float).reshape(3) if terminal else _bridge.host(self._eval(x, u, i)["lu"])[0][:3]

    def l_xx(self, x, u, i, terminal=False):
        return _bridge.host(self._eval(x, u, i, terminal)["lxx"])[0][np.ix_(self._IDX, self._IDX)]

    def l_ux(self, x, u, i, terminal=False):
        return np.zeros((self.action_size, self.state_size))

    def l_uu(self, x, u, i, terminal=False):
        return 2 * self._R


class AutoDiffCost(BaseCost):
    """Auto-differentiated Instantaneous Cost (traopt_cost.py:113-290) on torch.func: l(x, u, i) and
    l_terminal(x, i) take torch tensors and return a scalar tensor."""

    def __init__(self, l, l_terminal, state_size, action_size, device=None, **kwargs):  # noqa: E741
        from ._autodiff import Derivs
        self._state_size = state_size
        self._action_size = action_size
        self._d = Derivs(l, second=True, device=device)
        self._t = Derivs(lambda x, u, i: l_terminal(x, i), second=True, device=device)

    state_size = property(lambda self: self._state_size)
    action_size = property(lambda self: self._action_size)

    def _zu(self):
        return np.zeros(self._action_size)

    def l(self, x, u, i, terminal=False):  # noqa: E741
        return float(self._t.one("fn", x, self._zu(), i)) if terminal else float(self._d.one("fn", x, u, i))

    def l_x(self, x, u, i, terminal=False):
        return self._t.one("fx", x, self._zu(), i) if terminal else self._d.one("fx", x, u, i)

    def l_u(self, x, u, i, terminal=False):
        return np.zeros(self._action_size) if terminal else self._d.one("fu", x, u, i)  # traopt_cost.py:230-232

    def l_xx(self, x, u, i, terminal=False):
        return self._t.one("fxx", x, self._zu(), i) if terminal else self._d.one("fxx", x, u, i)

    def l_ux(self, x, u, i, terminal=False):
        if terminal:
            return np.zeros((self._action_size, self._state_size))
        return self._d.one("fux", x, u, i)

    def l_uu(self, x, u, i, terminal=False):
        if terminal:
            return np.zeros((self._action_size, self._action_size))
        return self._d.one("fuu", x, u, i)

    def batch(self, which, xs, us):
        """l / l_x / l_u / l_xx / l_ux / l_uu for all stage knots in one vmapped call."""
        return self._d.batch({"l": "fn", "l_x": "fx", "l_u": "fu", "l_xx": "fxx", "l_ux": "fux", "l_uu": "fuu"}[which],
                             xs, us)
