"""Dynamics plugins with the reference's interface (traoptlibrary/traopt_dynamics.py).

BaseDynamics :14-130, SE3Dynamics :629-898, RigidBodyDynamics :901-1206, DroneDynamics :1209-1530.
The per-knot methods evaluate on the GPU through tolg_eval_knot; the controllers never call them in
their loops (they hand the whole problem to the fused kernels)."""
import abc

import numpy as np

from . import _bridge


class BaseDynamics():
    """Dynamics Model (traopt_dynamics.py:14-130)."""

    @property
    @abc.abstractmethod
    def state_size(self):
        raise NotImplementedError

    @property
    @abc.abstractmethod
    def action_size(self):
        raise NotImplementedError

    @property
    @abc.abstractmethod
    def has_hessians(self):
        raise NotImplementedError

    @abc.abstractmethod
    def f(self, x, u, i):
        raise NotImplementedError

    @abc.abstractmethod
    def f_x(self, x, u, i):
        raise NotImplementedError

    @abc.abstractmethod
    def f_u(self, x, u, i):
        raise NotImplementedError

    @abc.abstractmethod
    def f_xx(self, x, u, i):
        raise NotImplementedError

    @abc.abstractmethod
    def f_ux(self, x, u, i):
        raise NotImplementedError

    @abc.abstractmethod
    def f_uu(self, x, u, i):
        raise NotImplementedError


class _RigidBodyOnSE3(BaseDynamics):
    _kind = "se3"

    def __init__(self, J, dt, integration_method="euler", state_size=(6, 6), action_size=6, hessians=False,
                 debug=None, **kwargs):
        self._state_size = state_size[0] + state_size[1]
        self._error_state_size = state_size[0]
        self._vel_state_size = state_size[1]
        self._action_size = action_size
        J = np.asarray(J, dtype=float)
        self._Ib = J[0:3, 0:3]
        self._m = J[4, 4]
        self._J = J
        self._Jinv = np.linalg.inv(J)
        self._dt = dt
        self._integration_method = integration_method
        if integration_method == "euler":
            pass
        elif integration_method == "rk4":
            raise ValueError("RK4 not implemented yet.")  # traopt_dynamics.py:676-678
        else:
            raise ValueError("Invalid integration method. Choose 'euler' or 'rk4'.")
        self._has_hessians = hessians
        self._debug = debug
        self._probe_solver = None

    state_size = property(lambda self: self._state_size)
    error_state_size = property(lambda self: self._error_state_size)
    vel_state_size = property(lambda self: self._vel_state_size)
    action_size = property(lambda self: self._action_size)
    has_hessians = property(lambda self: self._has_hessians)
    Ib = property(lambda self: self._Ib)
    m = property(lambda self: self._m)
    J = property(lambda self: self._J)
    Jinv = property(lambda self: self._Jinv)
    dt = property(lambda self: self._dt)

    def _probe(self):
        if self._probe_solver is None:
            self._probe_solver = _bridge.dynamics_probe(self._kind, self._J, self._dt)
        return self._probe_solver

    def _eval(self, x, u):
        q, xi = _bridge.split_state(x)
        return self._probe().eval_knot(0, q, xi, np.asarray(u, float).reshape(1, self._action_size))

    def f(self, x, u, i):
        """Next state [q (4,4), xi (6,)] (traopt_dynamics.py:763-800)."""
        r = self._eval(x, u)
        return [_bridge.host(r["f_q"])[0], _bridge.host(r["f_xi"])[0]]

    def fd_euler(self, x, u, i):
        return self.f(x, u, i)

    def f_x(self, x, u, i):
        """df/dx [12, 12] (traopt_dynamics.py:802-837), the reference's literal Jacobian."""
        return _bridge.host(self._eval(x, u)["Fx"])[0]

    def f_u(self, x, u, i):
        """df/du [12, m] (traopt_dynamics.py:839-850)."""
        return _bridge.host(self._eval(x, u)["Fu"])[0]

    def f_xx(self, x, u, i):
        raise NotImplementedError  # traopt_dynamics.py:863-866: never available for the exact models

    def f_ux(self, x, u, i):
        raise NotImplementedError

    def f_uu(self, x, u, i):
        raise NotImplementedError


class SE3Dynamics(_RigidBodyOnSE3):
    """Error-State SE(3) Dynamics Model (traopt_dynamics.py:629-898)."""
    _kind = "se3"


class RigidBodyDynamics(_RigidBodyOnSE3):
    """SE(3) Dynamics with Gravity (traopt_dynamics.py:901-1206)."""
    _kind = "rigidbody"

    def __init__(self, J, dt, integration_method="euler", state_size=(6, 6), action_size=6, hessians=False,
                 debug=None, **kwargs):
        super().__init__(J, dt, integration_method, state_size, action_size, hessians, debug, **kwargs)
        self._g = 9.8

    g = property(lambda self: self._g)


class DroneDynamics(_RigidBodyOnSE3):
    """Drone Dynamics: SE(3) dynamics with gravity and the 4 -> 6 input map (traopt_dynamics.py:1209-1530)."""
    _kind = "drone"

    def __init__(self, J, dt, integration_method="euler", state_size=(6, 6), action_size=4, hessians=False,
                 debug=None, **kwargs):
        super().__init__(J, dt, integration_method, state_size, action_size, hessians, debug, **kwargs)
        self._g = 9.8
        self._Pu = np.zeros((6, 4))
        self._Pu[0, 0] = self._Pu[1, 1] = self._Pu[2, 2] = self._Pu[5, 3] = 1.0

    g = property(lambda self: self._g)
    Pu = property(lambda self: self._Pu)


def _so3_state(x):
    """[SO3, SO3Tangent] (or [R (3,3), w (3,)]) -> embedded SE(3) state arrays."""
    q, xi = x
    R = q.rotation() if hasattr(q, "rotation") else np.asarray(q, float)
    w = xi.coeffs() if hasattr(xi, "coeffs") else np.asarray(xi, float)
    T = np.eye(4)
    T[:3, :3] = R
    return T.reshape(1, 4, 4), np.r_[np.asarray(w, float).reshape(3), 0.0, 0.0, 0.0].reshape(1, 6)


class SO3Dynamics(BaseDynamics):
    """Error-State SO(3) Dynamics Model (traopt_dynamics.py:275-418).  States are [SO3, SO3Tangent]."""

    def __init__(self, J, dt, integration_method="euler", state_size=(3, 3), action_size=3, hessians=False,
                 debug=None, **kwargs):
        self._state_size = state_size[0] + state_size[1]
        self._pos_state_size = state_size[0]
        self._vel_state_size = state_size[1]
        self._error_state_size = state_size[0]
        self._action_size = action_size
        self._J = np.asarray(J, dtype=float)
        self._Jinv = np.linalg.inv(self._J)
        self._dt = dt
        self._integration_method = integration_method
        if integration_method == "euler":
            pass
        elif integration_method == "rk4":
            raise ValueError("RK4 not implemented yet.")  # traopt_dynamics.py:318-320
        else:
            raise ValueError("Invalid integration method. Choose 'euler' or 'rk4'.")
        self._has_hessians = hessians
        self._debug = debug
        self._probe_solver = None

    state_size = property(lambda self: self._state_size)
    pos_state_size = property(lambda self: self._pos_state_size)
    vel_state_size = property(lambda self: self._vel_state_size)
    action_size = property(lambda self: self._action_size)
    has_hessians = property(lambda self: self._has_hessians)
    J = property(lambda self: self._J)
    Jinv = property(lambda self: self._Jinv)
    dt = property(lambda self: self._dt)

    def _probe(self):
        if self._probe_solver is None:
            J6 = np.eye(6)
            J6[:3, :3] = self._J
            self._probe_solver = _bridge.dynamics_probe("so3", J6, self._dt)
        return self._probe_solver

    def _eval(self, x, u):
        q, xi = _so3_state(x)
        return self._probe().eval_knot(0, q, xi, np.r_[np.asarray(u, float).reshape(3), 0, 0, 0].reshape(1, 6))

    def f(self, x, u, i):
        from .traopt_utilis import SO3, SO3Tangent
        r = self._eval(x, u)
        return [SO3.from_matrix(_bridge.host(r["f_q"])[0][:3, :3]), SO3Tangent(_bridge.host(r["f_xi"])[0][:3])]

    fd_euler = f

    def f_x(self, x, u, i):
        F = _bridge.host(self._eval(x, u)["Fx"])[0]
        idx = [0, 1, 2, 6, 7, 8]
        return F[np.ix_(idx, idx)]

    def f_u(self, x, u, i):
        F = _bridge.host(self._eval(x, u)["Fu"])[0]
        return F[np.ix_([0, 1, 2, 6, 7, 8], [0, 1, 2])]

    def f_xx(self, x, u, i):
        raise NotImplementedError

    def f_ux(self, x, u, i):
        raise NotImplementedError

    def f_uu(self, x, u, i):
        raise NotImplementedError


class Pendulum3dDyanmics(SO3Dynamics):
    """A dynamics model for 3d pendulum actuated by the pivot point (traopt_dynamics.py:421-626; the class
    name keeps the reference's spelling).  States are [SO3, SO3Tangent], the input is the pivot
    acceleration in R^3; f_u depends on the state and the lower-left block of f_x on the input."""

    def __init__(self, J, m, length, dt, integration_method="euler", state_size=(3, 3), action_size=3,
                 hessians=False, debug=None, **kwargs):
        super().__init__(J, dt, integration_method=integration_method, state_size=state_size,
                         action_size=action_size, hessians=hessians, debug=debug, **kwargs)
        self._m = m
        self._l = length
        self._g = 9.8  # traopt_dynamics.py:466

    m = property(lambda self: self._m)
    l = property(lambda self: self._l)  # noqa: E741  (the reference's property name)
    g = property(lambda self: self._g)

    def _probe(self):
        if self._probe_solver is None:
            J6 = np.eye(6)
            J6[:3, :3] = self._J
            self._probe_solver = _bridge.dynamics_probe("pendulum3d", J6, self._dt, self._m, self._l)
        return self._probe_solver


class AutoDiffDynamics(BaseDynamics):
    """Auto-differentiated Dynamics Model (traopt_dynamics.py:133-270), on torch.func where the reference
    uses jax: f(x, u, i) takes and returns torch tensors.  Config 1 of BASELINE.json (main_ddp.py) is
    plumbing around this class and the Euclidean iLQR."""

    def __init__(self, f, state_size, action_size, hessians=False, device=None, **kwargs):
        from ._autodiff import Derivs
        self._state_size = state_size
        self._action_size = action_size
        self._has_hessians = hessians
        self._d = Derivs(f, second=hessians, device=device)

    state_size = property(lambda self: self._state_size)
    action_size = property(lambda self: self._action_size)
    has_hessians = property(lambda self: self._has_hessians)

    def f(self, x, u, i):
        return self._d.one("fn", x, u, i)

    def f_x(self, x, u, i):
        return self._d.one("fx", x, u, i)

    def f_u(self, x, u, i):
        return self._d.one("fu", x, u, i)

    def f_xx(self, x, u, i):
        if not self._has_hessians:
            raise NotImplementedError
        return self._d.one("fxx", x, u, i)

    def f_ux(self, x, u, i):
        if not self._has_hessians:
            raise NotImplementedError
        return self._d.one("fux", x, u, i)

    def f_uu(self, x, u, i):
        if not self._has_hessians:
            raise NotImplementedError
        return self._d.one("fuu", x, u, i)

    def batch(self, which, xs, us):
        """f_x / f_u / f_xx / f_ux / f_uu for all knots of a rollout in one vmapped call."""
        return self._d.batch({"f_x": "fx", "f_u": "fu", "f_xx": "fxx", "f_ux": "fux", "f_uu": "fuu"}[which], xs, us)
