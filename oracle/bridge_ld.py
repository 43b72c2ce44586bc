"""ctypes bridge to oracle/libtolg_oracle_ld.so, the oracle in long double (tolg_oracle_ld.c) -- TEST INFRASTRUCTURE ONLY.

A referee, not a parity target: `fit_batch` takes the same (double) inputs as `oracle.bridge.fit_batch` and returns the
histories the 80-bit evaluation of the same statements produces, so that a rounding-level disagreement between the GPU
path and the fp64 oracle can be attributed (tools/parity_referee.py).  Only tools/ and tests/ import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtolg_oracle_ld.so")
LD = np.longdouble
assert np.finfo(LD).nmant >= 63, "long double is not the x87 extended type here: the referee would referee nothing"
_lp = C.POINTER(C.c_longdouble)
_ip = C.POINTER(C.c_int)


class Problem(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("m", C.c_int), ("N", C.c_int), ("dt", C.c_longdouble),
        ("J", C.c_longdouble * 36), ("Q", C.c_longdouble * 144), ("P", C.c_longdouble * 144), ("R", C.c_longdouble * 36),
        ("q_ref", _lp), ("xi_ref", _lp),
        ("al_on", C.c_int), ("al_lb", _lp), ("al_ub", _lp), ("al_lambda", _lp), ("al_imu", _lp),
        ("pend_mass", C.c_longdouble), ("pend_length", C.c_longdouble),
    ]


class Options(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("tol_grad", C.c_longdouble), ("tol_defect", C.c_longdouble),
                ("line_search", C.c_int), ("rollout_linear", C.c_int), ("max_reg", C.c_longdouble)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = [os.path.join(_HERE, f) for f in ("tolg_oracle.c", "tolg_oracle_ld.c")]
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "libtolg_oracle_ld.so"])
        _lib = C.CDLL(_SO)
    return _lib


def _c(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(LD))
    return a.reshape(shape) if shape is not None else a


def _p(a):
    return a.ctypes.data_as(_lp)


class OracleProblemLD:
    """The long-double image of an `oracle.bridge.OracleProblem` (same numbers, widened exactly)."""

    def __init__(self, op):
        c = op.c
        self.m, self.N = op.m, op.N
        self.q_ref, self.xi_ref = _c(op.q_ref), _c(op.xi_ref)
        p = Problem()
        p.kind, p.m, p.N, p.dt = c.kind, c.m, c.N, c.dt
        for name in ("J", "Q", "P", "R"):
            getattr(p, name)[:] = list(getattr(c, name))
        p.q_ref, p.xi_ref = _p(self.q_ref), _p(self.xi_ref)
        p.pend_mass, p.pend_length = c.pend_mass, c.pend_length
        p.al_on = 0
        if c.al_on:
            self.al = [_c(a) for a in op.al]
            p.al_on = 1
            p.al_lb, p.al_ub, p.al_lambda, p.al_imu = [_p(a) for a in self.al]
        self.c = p


def fit_batch(op, x0_q, x0_xi, us_init, mode="ms", max_iter=20, tol_grad=0.0, tol_defect=0.0, line_search=False,
              rollout="nonlinear", max_reg=1e10, threads=None):
    """`oracle.bridge.fit_batch` evaluated in long double; `op` is the fp64 OracleProblem.  Arrays come back as
    np.longdouble (cast to float64 for printing: the interesting part is the DIFFERENCE to the fp64 runs)."""
    prob = OracleProblemLD(op)
    B, N, m, K = x0_q.shape[0], prob.N, prob.m, max_iter
    o = Options(max_iter, tol_grad, tol_defect, int(line_search), int(rollout == "linear"), max_reg)
    x0_q = _c(x0_q, (B, 16)); x0_xi = _c(x0_xi, (B, 6)); us_init = _c(us_init, (B, N, m))
    z = lambda *s: np.zeros(s, dtype=LD)  # noqa: E731
    xs_q, xs_xi, us = z(B, N + 1, 4, 4), z(B, N + 1, 6), z(B, N, m)
    J_hist, grad_hist, defect_hist = np.full((B, K), np.nan, LD), np.full((B, K + 1), np.nan, LD), np.full((B, K + 1), np.nan, LD)
    iters, status, conv = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
    mu_hist = np.full((B, K), np.nan, LD)
    used = lib().tolg_oracle_fit_batch(int(mode == "ms"), C.byref(prob.c), C.byref(o), B, _p(x0_q), _p(x0_xi), _p(us_init),
                                       _p(xs_q), _p(xs_xi), _p(us), _p(J_hist), _p(grad_hist), _p(defect_hist),
                                       iters.ctypes.data_as(_ip), status.ctypes.data_as(_ip), conv.ctypes.data_as(_ip),
                                       int(threads or 0), _p(mu_hist))
    if used < 0:
        raise RuntimeError("long-double oracle fit_batch failed")
    return dict(xs_q=xs_q, xs_xi=xs_xi, us=us, J_hist=J_hist, grad_hist=grad_hist, defect_hist=defect_hist, iters=iters,
                status=status, converged=conv, mu_hist=mu_hist)


class History(C.Structure):
    _fields_ = [("J_hist", _lp), ("grad_hist", _lp), ("defect_hist", _lp), ("alpha_hist", _lp), ("mu_hist", _lp), ("J_lin", _lp),
                ("trial_J", _lp), ("n_trials", _ip), ("n_iters", C.c_int), ("converged", C.c_int), ("status", C.c_int)]


def fit(op, x0_q, x0_xi, us_init, mode="ms", max_iter=200, tol_grad=0.0, tol_defect=0.0, line_search=False,
        rollout="nonlinear", max_reg=1e10):
    """`oracle.bridge.fit` (one trajectory, with the cost of every line-search trial) in long double."""
    prob = OracleProblemLD(op)
    N, m, K = prob.N, prob.m, max_iter
    o = Options(max_iter, tol_grad, tol_defect, int(line_search), int(rollout == "linear"), max_reg)
    arrs = dict(J_hist=np.full(K, np.nan, LD), grad_hist=np.full(K + 1, np.nan, LD), defect_hist=np.full(K + 1, np.nan, LD),
                alpha_hist=np.full(K, np.nan, LD), mu_hist=np.full(K, np.nan, LD), J_lin=np.full(K + 1, np.nan, LD),
                trial_J=np.full((K, 20), np.nan, LD))
    n_trials = np.zeros(K, dtype=np.int32)
    h = History()
    for k, a in arrs.items():
        setattr(h, k, _p(a))
    h.n_trials = n_trials.ctypes.data_as(_ip)
    xs_q, xs_xi, us = np.zeros((N + 1, 4, 4), LD), np.zeros((N + 1, 6), LD), np.zeros((N, m), LD)
    x0_q = _c(x0_q, (16,)); x0_xi = _c(x0_xi, (6,)); us_init = _c(us_init, (N, m))
    fn = lib().tolg_oracle_ms_fit if mode == "ms" else lib().tolg_oracle_ss_fit
    rc = fn(C.byref(prob.c), C.byref(o), _p(x0_q), _p(x0_xi), _p(us_init), _p(xs_q), _p(xs_xi), _p(us), C.byref(h))
    if rc:
        raise RuntimeError("long-double oracle fit failed rc=%d" % rc)
    return dict(xs_q=xs_q, xs_xi=xs_xi, us=us, n_iters=h.n_iters, converged=bool(h.converged), status=h.status,
                n_trials=n_trials, **arrs)


def lin_backward(op, xs_q, xs_xi, us, ms=True, mu=1.0, delta=2.0, max_reg=1e10):
    """`oracle.bridge.lin_backward` (one linearisation + backward sweep of a given trajectory) in long double."""
    prob = OracleProblemLD(op)
    N, m = prob.N, prob.m
    xs_q = _c(xs_q, (N + 1, 16)); xs_xi = _c(xs_xi, (N + 1, 6)); us = _c(us, (N, m))
    z = lambda *s: np.zeros(s, dtype=LD)  # noqa: E731
    d, Fx, Lx, Lxx, k, K, Vx0, Vxx0 = z(N, 12), z(N, 12, 12), z(N + 1, 12), z(N + 1, 12, 12), z(N, m), z(N, m, 12), z(12), z(12, 12)
    J, g, mu_o, de_o = C.c_longdouble(), C.c_longdouble(), C.c_longdouble(), C.c_longdouble()
    rc = lib().tolg_oracle_lin_backward(C.byref(prob.c), int(ms), C.c_longdouble(mu), C.c_longdouble(delta), C.c_longdouble(max_reg),
                                        _p(xs_q), _p(xs_xi), _p(us), _p(d), _p(Fx), _p(Lx), _p(Lxx), _p(k), _p(K), _p(Vx0),
                                        _p(Vxx0), C.byref(J), C.byref(g), C.byref(mu_o), C.byref(de_o))
    if rc:
        raise RuntimeError("long-double oracle lin_backward rc=%d" % rc)
    return dict(d=d, Fx=Fx, Lx=Lx, Lxx=Lxx, k=k, K=K, Vx0=Vx0, Vxx0=Vxx0, J=J.value, grad=g.value, mu=mu_o.value, delta=de_o.value)
