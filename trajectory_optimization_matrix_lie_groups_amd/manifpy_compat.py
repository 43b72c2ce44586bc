"""Host-side stand-in for the slice of ``manifpy`` that the reference's library and scripts touch
(SURVEY.md App. D): SO3 / SO3Tangent / SE3 / SE3Tangent with rplus, lminus, rminus, inverse, act (optional
Jacobian out-arguments), exp / log, smallAdj, operators, in manif's conventions -- quaternion xyzw,
SE3 coefficients [t, q], SE3 tangent [v, w], right Jacobians throughout ("A micro Lie theory", Sola et al.).

This is NOT the hot path and not a fallback for it: scripts use these objects to build initial states and
to post-process solutions (errors, plots) a handful of elements at a time; the solvers never call it (the
same arithmetic runs batched on the device, csrc/tolg_lie.h).  ``install_as_manifpy()`` registers it
under the name ``manifpy`` when the real package is absent, so ``from manifpy import SE3, SO3Tangent``
in a reference script resolves.
"""
import sys

import numpy as np

_EPS = 1e-10  # manif's small-angle threshold


def _skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def _q_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def _so3_exp_q(w):
    th2 = float(w @ w)
    th = np.sqrt(th2)
    if th2 > _EPS:
        return np.r_[np.sin(th / 2) / th * w, np.cos(th / 2)]
    q = np.r_[0.5 * w, 1.0]
    return q / np.linalg.norm(q)


def _so3_log_q(q):
    v, w = q[:3], q[3]
    n2 = float(v @ v)
    if n2 > _EPS:
        n = np.sqrt(n2)
        # angle in (-pi, pi]: atan2 of the flipped quaternion when w < 0 (double cover)
        two_atan = 2 * np.arctan2(-n, -w) if w < 0 else 2 * np.arctan2(n, w)
        return two_atan / n * v
    return (2.0 / w) * v  # sign-aware small-angle branch


def _so3_ljac(w):
    th2 = float(w @ w)
    W = _skew(w)
    if th2 <= _EPS:
        return np.eye(3) + 0.5 * W
    th = np.sqrt(th2)
    return np.eye(3) + (1 - np.cos(th)) / th2 * W + (th - np.sin(th)) / (th2 * th) * W @ W


def _so3_ljacinv(w):
    th2 = float(w @ w)
    W = _skew(w)
    if th2 <= _EPS:
        return np.eye(3) - 0.5 * W
    th = np.sqrt(th2)
    return np.eye(3) - 0.5 * W + (1 / th2 - (1 + np.cos(th)) / (2 * th * np.sin(th))) * W @ W


def _se3_Q(v, w):
    """Barfoot's Q block of the SE(3) left Jacobian (manif SE3Tangent::fillQ)."""
    th2 = float(w @ w)
    V, W = _skew(v), _skew(w)
    WV, VW, WVW = W @ V, V @ W, W @ V @ W
    if th2 <= _EPS:
        A, B, C, D = 0.5, 1.0 / 6, -1.0 / 24, -1.0 / 60
    else:
        th = np.sqrt(th2)
        s, c = np.sin(th), np.cos(th)
        A = 0.5
        B = (th - s) / (th2 * th)
        C = (1 - th2 / 2 - c) / (th2 * th2)
        D = (C - 3 * (th - s - th2 * th / 6) / (th2 * th2 * th)) * 0.5
    return (A * V + B * (WV + VW + WVW) - C * (W @ WV + VW @ W - 3 * WVW) - D * (WVW @ W + W @ WVW))


def _fill(J, val):
    if J is not None:
        J[...] = val


# --------------------------------------------------------------------------------------------- SO(3)
class SO3Tangent:
    DoF, Dim = 3, 3

    def __init__(self, w=(0.0, 0.0, 0.0)):
        self._w = np.array(w, dtype=float).reshape(3)

    def coeffs(self):
        return self._w.copy()

    def hat(self):
        return _skew(self._w)

    def smallAdj(self):
        return _skew(self._w)

    def exp(self, J=None):
        _fill(J, self.rjac())
        return SO3(_so3_exp_q(self._w))

    def rjac(self):
        return _so3_ljac(-self._w)

    def ljac(self):
        return _so3_ljac(self._w)

    def rjacinv(self):
        return _so3_ljacinv(-self._w)

    def ljacinv(self):
        return _so3_ljacinv(self._w)

    def weightedNorm(self):
        return float(np.linalg.norm(self._w))

    def __mul__(self, s):
        return SO3Tangent(self._w * float(s))

    __rmul__ = __mul__

    def __truediv__(self, s):
        return SO3Tangent(self._w / float(s))

    def __neg__(self):
        return SO3Tangent(-self._w)

    def __add__(self, o):
        if isinstance(o, SO3):
            return o.lplus(self)
        return SO3Tangent(self._w + (o._w if isinstance(o, SO3Tangent) else np.asarray(o, float).reshape(3)))

    def __sub__(self, o):
        return SO3Tangent(self._w - (o._w if isinstance(o, SO3Tangent) else np.asarray(o, float).reshape(3)))

    def __repr__(self):
        return "SO3Tangent(%s)" % self._w


class SO3:
    DoF, Dim, RepSize = 3, 3, 4

    def __init__(self, *args, quaternion=None):
        if quaternion is None:
            if len(args) == 1:
                quaternion = args[0]
            elif len(args) == 4:
                quaternion = args
            elif len(args) == 0:
                quaternion = (0, 0, 0, 1.0)
            else:
                raise TypeError("SO3(quaternion xyzw) or SO3(x, y, z, w)")
        q = np.array(quaternion, dtype=float).reshape(4)
        self._q = q / np.linalg.norm(q)

    @classmethod
    def Identity(cls):
        return cls((0, 0, 0, 1.0))

    @classmethod
    def from_matrix(cls, R):
        from scipy.spatial.transform import Rotation
        return cls(Rotation.from_matrix(np.asarray(R, float)).as_quat())

    def coeffs(self):
        return self._q.copy()

    def quat(self):
        return self._q.copy()

    def rotation(self):
        return _q_to_R(self._q)

    def transform(self):
        T = np.eye(4)
        T[:3, :3] = self.rotation()
        return T

    def adj(self):
        return self.rotation()

    def inverse(self, J=None):
        _fill(J, -self.rotation())
        return SO3(np.r_[-self._q[:3], self._q[3]])

    def compose(self, o, J_a=None, J_b=None):
        _fill(J_a, o.rotation().T)
        _fill(J_b, np.eye(3))
        return SO3(_qmul(self._q, o._q))

    def log(self, J=None):
        t = SO3Tangent(_so3_log_q(self._q))
        _fill(J, t.rjacinv())
        return t

    def act(self, v, J_x=None, J_v=None):
        v = np.asarray(v, dtype=float).reshape(3)
        R = self.rotation()
        _fill(J_x, -R @ _skew(v))
        _fill(J_v, R)
        return R @ v

    def rplus(self, t, J_x=None, J_t=None):
        e = t.exp()
        _fill(J_x, e.rotation().T)
        _fill(J_t, t.rjac())
        return self.compose(e)

    plus = rplus

    def lplus(self, t, J_x=None, J_t=None):
        _fill(J_x, np.eye(3))
        _fill(J_t, self.rotation().T @ t.ljac())  # Ad(X)^-1 Jl(t)
        return t.exp().compose(self)

    def rminus(self, o, J_a=None, J_b=None):
        t = o.inverse().compose(self).log()
        _fill(J_a, t.rjacinv())
        _fill(J_b, -t.ljacinv())
        return t

    minus = rminus

    def lminus(self, o, J_a=None, J_b=None):
        t = self.compose(o.inverse()).log()
        Ji = t.rjacinv() @ o.rotation()
        _fill(J_a, Ji)
        _fill(J_b, -Ji)
        return t

    def between(self, o):
        return self.inverse().compose(o)

    def __mul__(self, o):
        return self.compose(o)

    def __add__(self, t):
        return self.rplus(t)

    def __sub__(self, o):
        return self.rminus(o)

    def __repr__(self):
        return "SO3(%s)" % self._q


# --------------------------------------------------------------------------------------------- SE(3)
class SE3Tangent:
    DoF, Dim = 6, 3

    def __init__(self, tau=(0.0,) * 6):
        self._c = np.array(tau, dtype=float).reshape(6)  # [v, w]

    def coeffs(self):
        return self._c.copy()

    def lin(self):
        return self._c[:3].copy()

    def ang(self):
        return self._c[3:].copy()

    def hat(self):
        M = np.zeros((4, 4))
        M[:3, :3] = _skew(self._c[3:])
        M[:3, 3] = self._c[:3]
        return M

    def smallAdj(self):
        W, V = _skew(self._c[3:]), _skew(self._c[:3])
        return np.block([[W, V], [np.zeros((3, 3)), W]])

    def exp(self, J=None):
        v, w = self._c[:3], self._c[3:]
        _fill(J, self.rjac())
        return SE3(_so3_ljac(w) @ v, _so3_exp_q(w))

    def ljac(self):
        v, w = self._c[:3], self._c[3:]
        Jl = _so3_ljac(w)
        return np.block([[Jl, _se3_Q(v, w)], [np.zeros((3, 3)), Jl]])

    def rjac(self):
        return SE3Tangent(-self._c).ljac()

    def ljacinv(self):
        v, w = self._c[:3], self._c[3:]
        Ji = _so3_ljacinv(w)
        return np.block([[Ji, -Ji @ _se3_Q(v, w) @ Ji], [np.zeros((3, 3)), Ji]])

    def rjacinv(self):
        return SE3Tangent(-self._c).ljacinv()

    def __mul__(self, s):
        return SE3Tangent(self._c * float(s))

    __rmul__ = __mul__

    def __truediv__(self, s):
        return SE3Tangent(self._c / float(s))

    def __neg__(self):
        return SE3Tangent(-self._c)

    def __add__(self, o):
        if isinstance(o, SE3):
            return o.lplus(self)
        return SE3Tangent(self._c + (o._c if isinstance(o, SE3Tangent) else np.asarray(o, float).reshape(6)))

    def __sub__(self, o):
        return SE3Tangent(self._c - (o._c if isinstance(o, SE3Tangent) else np.asarray(o, float).reshape(6)))

    def __repr__(self):
        return "SE3Tangent(%s)" % self._c


class SE3:
    DoF, Dim, RepSize = 6, 3, 7

    def __init__(self, *args, position=None, quaternion=None):
        if position is None and quaternion is None:
            if len(args) == 2:
                position, quaternion = args
            elif len(args) == 7:
                position, quaternion = args[:3], args[3:]
            elif len(args) == 1:
                position, quaternion = np.asarray(args[0], float)[:3], np.asarray(args[0], float)[3:]
            elif len(args) == 0:
                position, quaternion = (0, 0, 0), (0, 0, 0, 1.0)
            else:
                raise TypeError("SE3(position=, quaternion=xyzw)")
        self._t = np.array(position, dtype=float).reshape(3)
        q = np.array(quaternion, dtype=float).reshape(4)
        self._q = q / np.linalg.norm(q)

    @classmethod
    def Identity(cls):
        return cls((0, 0, 0), (0, 0, 0, 1.0))

    def coeffs(self):
        return np.r_[self._t, self._q]

    def quat(self):
        return self._q.copy()

    def rotation(self):
        return _q_to_R(self._q)

    def translation(self):
        return self._t.copy()

    def transform(self):
        T = np.eye(4)
        T[:3, :3] = self.rotation()
        T[:3, 3] = self._t
        return T

    def adj(self):
        R = self.rotation()
        return np.block([[R, _skew(self._t) @ R], [np.zeros((3, 3)), R]])

    def inverse(self, J=None):
        _fill(J, -self.adj())
        qi = np.r_[-self._q[:3], self._q[3]]
        return SE3(-(_q_to_R(qi) @ self._t), qi)

    def compose(self, o, J_a=None, J_b=None):
        _fill(J_a, o.inverse().adj())
        _fill(J_b, np.eye(6))
        return SE3(self._t + self.rotation() @ o._t, _qmul(self._q, o._q))

    def log(self, J=None):
        w = _so3_log_q(self._q)
        t = SE3Tangent(np.r_[_so3_ljacinv(w) @ self._t, w])
        _fill(J, t.rjacinv())
        return t

    def act(self, v, J_x=None, J_v=None):
        v = np.asarray(v, dtype=float).reshape(3)
        R = self.rotation()
        _fill(J_x, np.hstack([R, -R @ _skew(v)]))
        _fill(J_v, R)
        return R @ v + self._t

    def rplus(self, t, J_x=None, J_t=None):
        e = t.exp()
        _fill(J_x, e.inverse().adj())
        _fill(J_t, t.rjac())
        return self.compose(e)

    plus = rplus

    def lplus(self, t, J_x=None, J_t=None):
        _fill(J_x, np.eye(6))
        _fill(J_t, self.inverse().adj() @ t.ljac())
        return t.exp().compose(self)

    def rminus(self, o, J_a=None, J_b=None):
        t = o.inverse().compose(self).log()
        _fill(J_a, t.rjacinv())
        _fill(J_b, -t.ljacinv())
        return t

    minus = rminus

    def lminus(self, o, J_a=None, J_b=None):
        t = self.compose(o.inverse()).log()
        Ji = t.rjacinv() @ o.adj()
        _fill(J_a, Ji)
        _fill(J_b, -Ji)
        return t

    def between(self, o):
        return self.inverse().compose(o)

    def __mul__(self, o):
        return self.compose(o)

    def __add__(self, t):
        return self.rplus(t)

    def __sub__(self, o):
        return self.rminus(o)

    def __repr__(self):
        return "SE3(t=%s, q=%s)" % (self._t, self._q)


def install_as_manifpy(force=False):
    """Make ``import manifpy`` resolve to this module if the real package is not installed."""
    if not force:
        try:
            import manifpy  # noqa: F401
            return sys.modules["manifpy"]
        except ImportError:
            pass
    sys.modules["manifpy"] = sys.modules[__name__]
    return sys.modules[__name__]
