"""Formula-free checks of the oracle's Lie primitives: scipy expm/logm and
the identity Jl(tau) = expm([[ad tau, I],[0,0]])[:6, 6:]."""
import numpy as np
import pytest
from scipy.linalg import expm, logm

from oracle import bridge as ob


def skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def hat(xi):
    M = np.zeros((4, 4))
    M[:3, :3] = skew(xi[:3])
    M[:3, 3] = xi[3:]
    return M


def ad(xi):
    A = np.zeros((6, 6))
    A[:3, :3] = skew(xi[:3])
    A[3:, :3] = skew(xi[3:])
    A[3:, 3:] = skew(xi[:3])
    return A


def Jl_ref(tau):
    M = np.zeros((12, 12))
    M[:6, :6] = ad(tau)
    M[:6, 6:] = np.eye(6)
    return expm(M)[:6, 6:]


TWISTS = [np.random.default_rng(s).normal(size=6) * sc for s, sc in enumerate([0.05, 0.3, 1.0, 1.0, 2.0, 2.5, 1e-3, 1e-4])]


@pytest.mark.parametrize("tau", TWISTS)
def test_exp_log_match_scipy(tau):
    T = ob.se3_exp(tau)
    np.testing.assert_allclose(T, expm(hat(tau)), atol=2e-14)
    if np.linalg.norm(tau[:3]) < np.pi:
        np.testing.assert_allclose(ob.se3_log(T), tau, atol=1e-12)
    else:  # beyond the cut Log returns the principal value: compare through Exp
        np.testing.assert_allclose(ob.se3_exp(ob.se3_log(T)), T, atol=1e-12)
    if 1e-2 < np.linalg.norm(tau[:3]) < np.pi:
        L = np.real(logm(T))
        np.testing.assert_allclose(ob.se3_log(T), np.r_[L[2, 1], L[0, 2], L[1, 0], L[:3, 3]], atol=1e-11)


@pytest.mark.parametrize("tau", TWISTS)
def test_jacobians_match_expm_identity(tau):
    np.testing.assert_allclose(ob.se3_ljac(tau), Jl_ref(tau), atol=5e-13)
    np.testing.assert_allclose(ob.se3_rjac(tau), Jl_ref(-tau), atol=5e-13)
    np.testing.assert_allclose(ob.se3_rjacinv(tau), np.linalg.inv(Jl_ref(-tau)), atol=5e-11)
    T = ob.se3_exp(tau)
    np.testing.assert_allclose(ob.se3_adj(T), expm(ad(tau)), atol=1e-13)


def test_small_angle_branches_are_continuous():
    d = np.array([0.3, -0.2, 0.1])
    for th in [0.0, 1e-9, 9e-6, 1.1e-5, 1e-4]:
        tau = np.r_[th * np.array([0.6, 0.0, 0.8]), d]
        np.testing.assert_allclose(ob.se3_exp(tau), expm(hat(tau)), atol=1e-14)
        np.testing.assert_allclose(ob.se3_ljac(tau), Jl_ref(tau), atol=1e-10)
        np.testing.assert_allclose(ob.se3_log(ob.se3_exp(tau)), tau, atol=1e-13)


def test_log_near_pi_and_negative_w():
    for th in [3.0, np.pi - 1e-6, 3.5, 5.0]:
        w = th * np.array([1.0, 2.0, -2.0]) / 3.0
        tau = np.r_[w, 0.1, 0.2, 0.3]
        T = ob.se3_exp(tau)
        back = ob.se3_log(T)
        # Log returns the principal value; compare through Exp
        np.testing.assert_allclose(ob.se3_exp(back), T, atol=1e-12)
        assert np.linalg.norm(back[:3]) <= np.pi + 1e-12


def test_project_reorthonormalises():
    rng = np.random.default_rng(3)
    T = ob.se3_exp(rng.normal(size=6))
    Tn = T.copy()
    Tn[:3, :3] += 1e-9 * rng.normal(size=(3, 3))
    P = ob.project(Tn)
    assert np.abs(P[:3, :3].T @ P[:3, :3] - np.eye(3)).max() < 1e-15
    np.testing.assert_allclose(P, T, atol=5e-9)


def test_lminus_rminus_jacobians_by_finite_difference():
    rng = np.random.default_rng(5)
    A = ob.se3_exp(rng.normal(size=6))
    B = ob.se3_exp(rng.normal(size=6))
    e, J = ob.lminus(A, B)
    np.testing.assert_allclose(e, ob.se3_log(A @ np.linalg.inv(B)), atol=1e-12)
    np.testing.assert_allclose(ob.rminus(A, B), ob.se3_log(np.linalg.inv(B) @ A), atol=1e-12)
    eps = 1e-6
    FD = np.zeros((6, 6))
    for j in range(6):
        d = np.zeros(6)
        d[j] = eps
        FD[:, j] = (ob.lminus(A @ ob.se3_exp(d), B)[0] - e) / eps
    np.testing.assert_allclose(J, FD, atol=5e-6)
