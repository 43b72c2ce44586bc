"""Shared plumbing of the mirror classes: lazy probe solvers and array conversion."""
import numpy as np

from ..solver import BatchedTrackingILQR, TrackingProblem

_IDENT_REF = np.repeat(np.eye(4)[None], 2, 0)
_ZERO_XI = np.zeros((2, 6))


def kind_of_action_size(m):
    return "drone" if int(m) == 4 else "se3"


def dynamics_probe(kind, J, dt, pend_mass=0.0, pend_length=0.0):
    """One-knot solver used for dynamics.f / f_x / f_u (the cost terms are zero weights)."""
    m = 4 if kind == "drone" else 6
    prob = TrackingProblem(kind, np.asarray(J, float), float(dt), np.zeros((12, 12)), np.zeros((m, m)),
                           np.zeros((12, 12)), _IDENT_REF, _ZERO_XI, float(pend_mass), float(pend_length))
    return BatchedTrackingILQR(prob, 1)


def cost_probe(Q, R, P, q_ref, xi_ref, action_size):
    """Solver used for cost.l / l_x / l_xx / l_u / l_uu / _err (unit inertia: dynamics outputs unused)."""
    prob = TrackingProblem(kind_of_action_size(action_size), np.eye(6), 1.0, np.asarray(Q, float), np.asarray(R, float),
                           np.asarray(P, float), np.asarray(q_ref, float), np.asarray(xi_ref, float))
    return BatchedTrackingILQR(prob, 1)


def split_state(x):
    q, xi = x
    return np.asarray(q, float).reshape(1, 4, 4), np.asarray(xi, float).reshape(1, 6)


def host(t):
    return t.detach().cpu().numpy()
