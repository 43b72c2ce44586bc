"""manifpy stand-in against the oracle's SE(3) element functions -- which are
themselves pinned end-to-end by the reference's recorded runs (tests/test_oracle_golden.py) -- and
against finite differences for the Jacobian out-arguments of every operation the library calls."""
import numpy as np
import pytest

from oracle import bridge as ob
from trajectory_optimization_matrix_lie_groups_amd import manifpy_compat as mc
from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary.traopt_utilis import Jmnf2J, SE32manifSE3, manifse32se3, se32manifse3

RNG = np.random.default_rng(11)


def _rand_se3(scale=1.0):
    return mc.SE3Tangent(RNG.normal(size=6) * scale).exp()


def _wv(tau_vw):  # manif [v, w] -> reference twist order [w, v]
    return np.r_[tau_vw[3:], tau_vw[:3]]


def test_exp_log_jacobians_match_oracle():
    for scale in (1.0, 1e-7, 2.5):
        t = mc.SE3Tangent(RNG.normal(size=6) * scale)
        X = t.exp()
        assert np.abs(X.transform() - ob.se3_exp(_wv(t.coeffs()))).max() < 1e-14
        assert np.abs(_wv(X.log().coeffs()) - ob.se3_log(X.transform())).max() < 1e-12
        assert np.abs(Jmnf2J(t.rjac()) - ob.se3_rjac(_wv(t.coeffs()))).max() < 1e-13
        assert np.abs(Jmnf2J(t.ljac()) - ob.se3_ljac(_wv(t.coeffs()))).max() < 1e-13
        assert np.abs(Jmnf2J(t.rjacinv()) - ob.se3_rjacinv(_wv(t.coeffs()))).max() < 1e-11
        assert np.abs(Jmnf2J(X.adj()) - ob.se3_adj(X.transform())).max() < 1e-14
        assert np.abs(t.rjac() @ t.rjacinv() - np.eye(6)).max() < 1e-12


def test_lminus_rminus_rplus_match_oracle_and_reference_call_pattern():
    A, B = _rand_se3(), _rand_se3()
    J = np.empty((6, 6))
    e = A.lminus(B, J)  # traopt_cost.py:668
    eo, Jo = ob.lminus(A.transform(), B.transform())
    assert np.abs(manifse32se3(e) - eo).max() < 1e-13 and np.abs(Jmnf2J(J) - Jo).max() < 1e-11
    assert np.abs(manifse32se3(A - B) - ob.rminus(A.transform(), B.transform())).max() < 1e-13
    xi = RNG.normal(size=6)
    Jx, Jt = np.empty((6, 6)), np.empty((6, 6))
    Xn = A.rplus(se32manifse3(xi) * 0.05, Jx, Jt)  # traopt_dynamics.py:821-826
    assert np.abs(Xn.transform() - A.transform() @ ob.se3_exp(xi * 0.05)).max() < 1e-14
    assert np.abs(Jmnf2J(Jt) - ob.se3_rjac(xi * 0.05)).max() < 1e-13
    assert np.abs(Jmnf2J(Jx) - ob.se3_adj(np.linalg.inv(ob.se3_exp(xi * 0.05)))).max() < 1e-13
    assert np.abs((A + se32manifse3(xi)).transform() - A.transform() @ ob.se3_exp(xi)).max() < 1e-14
    assert np.abs((A * B).transform() - A.transform() @ B.transform()).max() < 1e-14
    assert np.abs(SE32manifSE3(A.transform()).coeffs() - A.coeffs()).max() < 1e-14 or \
        np.abs(SE32manifSE3(A.transform()).coeffs()[3:] + A.coeffs()[3:]).max() < 1e-14


def _fd(fun, n, h=1e-6):
    cols = []
    for k in range(n):
        d = np.zeros(n); d[k] = h
        cols.append((fun(d) - fun(-d)) / (2 * h))
    return np.stack(cols, axis=1)


@pytest.mark.parametrize("G,T", [(mc.SE3, mc.SE3Tangent), (mc.SO3, mc.SO3Tangent)])
def test_jacobian_out_arguments_are_right_jacobians(G, T):
    n = G.DoF
    X, Y = T(RNG.normal(size=n)).exp(), T(RNG.normal(size=n)).exp()
    t = T(RNG.normal(size=n) * 0.7)
    v = RNG.normal(size=3)

    def chk(out, jac_x, fx, jac_y=None, fy=None):
        assert np.abs(jac_x - _fd(fx, n)).max() < 1e-7
        if jac_y is not None:
            assert np.abs(jac_y - _fd(fy, jac_y.shape[1])).max() < 1e-7

    # inverse: d (X+d)^-1 (-) X^-1
    J = np.empty((n, n)); Xi = X.inverse(J)
    chk(Xi, J, lambda d: (X + T(d)).inverse().rminus(Xi).coeffs())
    # rplus
    Ja, Jb = np.empty((n, n)), np.empty((n, n)); Z = X.rplus(t, Ja, Jb)
    chk(Z, Ja, lambda d: (X + T(d)).rplus(t).rminus(Z).coeffs(), Jb, lambda d: X.rplus(t + T(d)).rminus(Z).coeffs())
    # rminus / lminus
    Ja, Jb = np.empty((n, n)), np.empty((n, n)); e = X.rminus(Y, Ja, Jb)
    chk(e, Ja, lambda d: (X + T(d)).rminus(Y).coeffs() - e.coeffs(), Jb, lambda d: X.rminus(Y + T(d)).coeffs() - e.coeffs())
    Ja, Jb = np.empty((n, n)), np.empty((n, n)); e = X.lminus(Y, Ja, Jb)
    chk(e, Ja, lambda d: (X + T(d)).lminus(Y).coeffs() - e.coeffs(), Jb, lambda d: X.lminus(Y + T(d)).coeffs() - e.coeffs())
    # act (traopt_dynamics.py:574-584)
    Ja, Jb = np.empty((3, n)), np.empty((3, 3)); p = X.act(v, Ja, Jb)
    chk(p, Ja, lambda d: (X + T(d)).act(v) - p, Jb, lambda d: X.act(v + d) - p)
    # log / exp
    J = np.empty((n, n)); lg = X.log(J)
    chk(lg, J, lambda d: (X + T(d)).log().coeffs() - lg.coeffs())
    J = np.empty((n, n)); ex = t.exp(J)
    chk(ex, J, lambda d: (t + T(d)).exp().rminus(ex).coeffs())


def test_so3_surface_and_operators():
    w = mc.SO3Tangent([0.3, -1.1, 0.5])
    R = w.exp()
    from scipy.spatial.transform import Rotation
    assert np.abs(R.rotation() - Rotation.from_rotvec(w.coeffs()).as_matrix()).max() < 1e-15
    assert np.abs(R.log().coeffs() - w.coeffs()).max() < 1e-14
    assert np.abs(w.smallAdj() @ np.array([1.0, 2, 3]) - np.cross(w.coeffs(), [1.0, 2, 3])).max() < 1e-15
    assert mc.SO3.DoF == 3 and np.allclose((w * 2).coeffs(), (2 * w).coeffs())
    assert np.abs(mc.SO3.Identity().rotation() - np.eye(3)).max() == 0
    assert np.abs((R * R.inverse()).rotation() - np.eye(3)).max() < 1e-15
    q = Rotation.from_euler("xy", [10.0, 45.0], degrees=True).as_quat()
    assert np.abs(mc.SO3(q).rotation() - Rotation.from_quat(q).as_matrix()).max() < 1e-15  # main_pendulum3d...:55
    # rotation by more than pi comes back as the short way round
    big = mc.SO3Tangent([0, 0, 3.5]).exp().log().coeffs()
    assert abs(big[2] - (3.5 - 2 * np.pi)) < 1e-14
    se = mc.SE3(position=np.array([1.0, 2, 3]), quaternion=q)  # benchmark_SE3_tracking.py:67-70
    assert np.abs(se.transform()[:3, 3] - [1, 2, 3]).max() == 0 and se.coeffs().shape == (7,)
    xi = mc.SE3Tangent([1, 2, 3, 0.1, 0.2, 0.3])
    assert np.abs(xi.smallAdj()[:3, 3:] - mc._skew(np.array([1.0, 2, 3]))).max() == 0


def test_install_as_manifpy():
    import sys
    had = sys.modules.get("manifpy")
    try:
        mod = mc.install_as_manifpy(force=True)
        from manifpy import SE3, SO3Tangent  # noqa: F401
        assert mod is mc and SE3 is mc.SE3
    finally:
        if had is None:
            sys.modules.pop("manifpy", None)
        else:
            sys.modules["manifpy"] = had
