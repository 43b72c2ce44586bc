"""Host-side checks of the traoptlibrary mirror that need no GPU: names, signatures, helpers."""
import inspect

import numpy as np
import pytest

import trajectory_optimization_matrix_lie_groups_amd as pkg


def test_install_as_traoptlibrary_exposes_reference_names():
    pkg.install_as_traoptlibrary()
    from traoptlibrary.traopt_controller import iLQR_Tracking_SE3, iLQR_Tracking_SE3_MS, AL_iLQR_Tracking_SE3_MS  # noqa
    from traoptlibrary.traopt_dynamics import SE3Dynamics, DroneDynamics, RigidBodyDynamics, BaseDynamics  # noqa
    from traoptlibrary.traopt_cost import (SE3TrackingQuadraticGaussNewtonCost,  # noqa
                                           ErrorStateSE3TrackingQuadraticGaussNewtonCost, ALConstrainedCost, BaseCost)
    from traoptlibrary.traopt_constraints import InputConstraint, BaseConstraint  # noqa
    from traoptlibrary.traopt_utilis import skew, se3_hat, se3_vee, adjoint, coadjoint, SE32manifSE3, is_pos_def  # noqa
    import traoptlibrary.traopt_baseline as tb
    with pytest.raises(NotImplementedError):
        tb.EmbeddedEuclideanSU2_SE3()
    assert ErrorStateSE3TrackingQuadraticGaussNewtonCost is SE3TrackingQuadraticGaussNewtonCost


def test_constructor_signatures_match_reference():
    """Argument names and defaults of SURVEY.md §8b (reference file:line in the mirror docstrings)."""
    from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary import traopt_controller as tc
    from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary import traopt_dynamics as td
    from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary import traopt_cost as tcost
    sig = lambda f: [(p.name, p.default) for p in inspect.signature(f).parameters.values()][1:]  # noqa: E731
    E = inspect.Parameter.empty
    assert sig(td.SE3Dynamics.__init__)[:7] == [("J", E), ("dt", E), ("integration_method", "euler"), ("state_size", (6, 6)),
                                                ("action_size", 6), ("hessians", False), ("debug", None)]
    assert dict(sig(td.DroneDynamics.__init__))["action_size"] == 4
    assert sig(tcost.SE3TrackingQuadraticGaussNewtonCost.__init__)[:7] == [
        ("Q", E), ("R", E), ("P", E), ("q_ref", E), ("xi_ref", E), ("state_size", (6, 6)), ("action_size", 6)]
    assert sig(tc.iLQR_Tracking_SE3.__init__) == [("dynamics", E), ("cost", E), ("N", E), ("max_reg", 1e10),
                                                  ("hessians", False), ("rollout", "linear"), ("debug", None)]
    assert sig(tc.iLQR_Tracking_SE3_MS.__init__) == [("dynamics", E), ("cost", E), ("N", E), ("q_ref", E), ("xi_ref", E),
                                                     ("max_reg", 1e10), ("hessians", False), ("line_search", False),
                                                     ("rollout", "linear"), ("debug", None)]
    assert sig(tc.iLQR_Tracking_SE3.fit) == [("x0", E), ("us_init", E), ("n_iterations", 100), ("tol_J", 1e-6),
                                             ("tol_grad_norm", 1e-3), ("on_iteration", None)]
    assert sig(tc.iLQR_Tracking_SE3_MS.fit) == [("x0", E), ("us_init", E), ("n_iterations", 100), ("tol_J", 1e-6),
                                                ("tol_grad_norm", 1e-6), ("tol_d_norm", 1e-6), ("on_iteration", None)]
    assert sig(tc.AL_iLQR_Tracking_SE3_MS.__init__)[:7] == [("dynamics", E), ("cost", E), ("constraints", E), ("N", E),
                                                            ("q_ref", E), ("xi_ref", E), ("mu_scale", 10.)]


def test_utilis_conventions():
    from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary import traopt_utilis as tu
    xi = np.arange(1.0, 7.0)
    H = tu.se3_hat(xi)
    np.testing.assert_array_equal(tu.se3_vee(H), xi)           # [omega, v] order
    np.testing.assert_array_equal(H[:3, 3], xi[3:])
    A = tu.adjoint(xi)
    np.testing.assert_array_equal(A[:3, :3], tu.skew(xi[:3])); np.testing.assert_array_equal(A[3:, :3], tu.skew(xi[3:]))
    np.testing.assert_array_equal(tu.coadjoint(xi), A.T)
    with pytest.raises(ValueError):
        tu.skew(np.zeros(4))
    J = np.arange(36.0).reshape(6, 6)
    np.testing.assert_array_equal(tu.Jmnf2J(tu.Jmnf2J(J)), J)
    assert tu.is_pos_def(np.eye(3)) and not tu.is_pos_def(-np.eye(3)) and not tu.is_pos_def(np.array([[1.0, 2], [0, 1]]))
    T = tu.SE3(position=[1, 2, 3], quaternion=[0, 0, np.sin(0.3), np.cos(0.3)]).transform()
    np.testing.assert_allclose(tu.manifSE32SE3(tu.SE32manifSE3(T)), T, atol=1e-15)


def test_input_constraint_values():
    from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary.traopt_constraints import InputConstraint
    c = InputConstraint(-2 * np.ones(6), 3 * np.ones(6))
    u = np.array([0, 1, -3, 4, 0, 0.0])
    np.testing.assert_array_equal(c.g(None, u, 0), np.r_[-2 - u, u - 3])
    assert c.g(None, None, 5, terminal=True).shape == (12,) and not c.g(None, None, 5, terminal=True).any()
    assert c.g_x(None, u, 0).shape == (12, 12) and c.g_u(None, u, 0).shape == (12, 6)
    np.testing.assert_array_equal(c.g_u(None, u, 0), np.vstack([-np.eye(6), np.eye(6)]))


def test_dynamics_argument_errors():
    from trajectory_optimization_matrix_lie_groups_amd.traoptlibrary.traopt_dynamics import SE3Dynamics
    with pytest.raises(ValueError, match="RK4 not implemented"):
        SE3Dynamics(np.eye(6), 0.01, integration_method="rk4")
    with pytest.raises(ValueError, match="Invalid integration method"):
        SE3Dynamics(np.eye(6), 0.01, integration_method="heun")
