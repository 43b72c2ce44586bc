"""Constraints with the reference's interface (traoptlibrary/traopt_constraints.py:5-169)."""
import abc

import numpy as np


class BaseConstraint():
    """traopt_constraints.py:5-63"""

    @abc.abstractmethod
    def g(self, x, u, i, terminal=False, *args, **kwargs):
        raise NotImplementedError

    @abc.abstractmethod
    def g_x(self, x, u, i, terminal=False, *args, **kwargs):
        raise NotImplementedError

    @abc.abstractmethod
    def g_u(self, x, u, i, terminal=False, *args, **kwargs):
        raise NotImplementedError


class InputConstraint(BaseConstraint):
    """Box input constraint g = [lb - u; u - ub] <= 0 (traopt_constraints.py:66-169).  The values are
    trivial affine maps of u (host code in the reference as well); the augmented-Lagrangian terms
    built from them run on the device (tolg_set_al / tolg_al_update)."""

    def __init__(self, input_lb, input_ub, state_size=(6, 6), action_size=6):
        self._state_size = state_size[0] + state_size[1]
        self._error_state_size = state_size[0]
        self._vel_state_size = state_size[1]
        self._action_size = action_size
        self._lb = input_lb
        self._ub = input_ub
        self._constr_size = 2 * action_size

    lb = property(lambda self: self._lb)
    ub = property(lambda self: self._ub)
    constr_size = property(lambda self: self._constr_size)
    state_size = property(lambda self: self._state_size)
    error_state_size = property(lambda self: self._error_state_size)
    vel_state_size = property(lambda self: self._vel_state_size)
    action_size = property(lambda self: self._action_size)

    def g(self, x, u, i, terminal=False, *args, **kwargs):
        if terminal:
            return np.zeros((self._constr_size,))
        return np.concatenate([self.lb - u, u - self.ub])

    def g_x(self, x, u, i, terminal=False, *args, **kwargs):
        return np.zeros([self.constr_size, self.state_size])

    def g_u(self, x, u, i, terminal=False, *args, **kwargs):
        if terminal:
            return np.zeros([self.constr_size, self.action_size])
        return np.vstack([-1 * np.identity(self.action_size), np.identity(self.action_size)])
