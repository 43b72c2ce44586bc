"""A caller with the shape of the reference's benchmark scripts (benchmark_SE3_tracking.py:1-217,
benchmark_drone_racing_tracking.py:48-216), written against the reference's import names:

    from traoptlibrary.traopt_dynamics import DroneDynamics ...      (-> install_as_traoptlibrary())
    from manifpy import SE3                                          (-> install_as_manifpy(), if manifpy is absent)

It builds the problem from a reference-trajectory file, the initial pose with `SE3(position=, quaternion=).transform()`,
runs the multiple-shooting solver with the 15-argument callback and the single-shooting solver with the
12-argument one (both callbacks fill the history lists, as in the reference), and saves the result dict in the
reference's layout (NumPy archive instead of a pickle).  tests/test_gpu_dropin.py runs it on the GPU.
"""
import sys

import numpy as np
from scipy.spatial.transform import Rotation

import trajectory_optimization_matrix_lie_groups_amd as tolg
from trajectory_optimization_matrix_lie_groups_amd import manifpy_compat, results_io

tolg.install_as_traoptlibrary()
manifpy_compat.install_as_manifpy()

from manifpy import SE3  # noqa: E402
from traoptlibrary.traopt_controller import iLQR_Tracking_SE3, iLQR_Tracking_SE3_MS  # noqa: E402
from traoptlibrary.traopt_cost import SE3TrackingQuadraticGaussNewtonCost  # noqa: E402
from traoptlibrary.traopt_dynamics import DroneDynamics, SE3Dynamics  # noqa: E402
from traoptlibrary.traopt_utilis import SE32manifSE3, se32manifse3  # noqa: E402


def on_iteration_ms_se3(iteration_count, xs, us, J_opt, accepted, converged, defect_norm, grad_wrt_input_norm,
                        alpha, mu, J_hist, xs_hist, us_hist, grad_hist, defect_hist):
    J_hist.append(J_opt)
    xs_hist.append(xs.copy())
    us_hist.append(us.copy())
    grad_hist.append(np.copy(grad_wrt_input_norm))
    defect_hist.append(defect_norm)
    info = "converged" if converged else ("accepted" if accepted else "failed")
    print("Iteration", iteration_count, info, J_opt, defect_norm, grad_wrt_input_norm, alpha, mu)


def on_iteration_ss_se3(iteration_count, xs, us, J_opt, accepted, converged, grad_wrt_input_norm, alpha, mu, J_hist,
                        xs_hist, us_hist):
    J_hist.append(J_opt)
    xs_hist.append(xs.copy())
    us_hist.append(us.copy())
    info = "converged" if converged else ("accepted" if accepted else "failed")
    print("Iteration", iteration_count, info, J_opt, grad_wrt_input_norm, alpha, mu)


def err_dyn(xk, xk1, dt):
    """Kinematic consistency of two consecutive knots, as the benchmarks plot it (benchmark_SE3_tracking.py:92-100)."""
    Xk_sim = SE32manifSE3(xk[0]).rplus(se32manifse3(xk[1]) * dt).transform()
    return np.linalg.norm(Xk_sim - xk1[0])


def main(problem_file, model="drone", max_iterations=200, tol=1e-12, save_to=None):
    with np.load(problem_file) as g:  # q_ref, xi_ref, dt (+ the weights of the recorded run)
        q_ref, xi_ref, dt = g["q_ref"], g["xi_ref"], float(g["dt"])
        J, Q, P, R = g["J"], g["Q"], g["P"], g["R"]
        xi0 = g["xi0"]
        if "position0" in g.files:
            position, quaternion = g["position0"], g["quaternion0"]
        else:  # the recorded problems store the initial pose as a matrix: back to what the scripts pass to manif
            position, quaternion = g["q0"][:3, 3], Rotation.from_matrix(g["q0"][:3, :3]).as_quat()
    N = q_ref.shape[0] - 1
    q0 = SE3(position=position, quaternion=quaternion).transform()
    x0 = [q0, xi0]
    action_size = 4 if model == "drone" else 6
    dynamics = (DroneDynamics if model == "drone" else SE3Dynamics)(J, dt, hessians=False)
    cost = SE3TrackingQuadraticGaussNewtonCost(Q, R, P, q_ref, xi_ref, action_size=action_size)
    us_init = np.zeros((N, action_size))
    ilqr_ms = iLQR_Tracking_SE3_MS(dynamics, cost, N, q_ref, xi_ref, hessians=False, line_search=False, rollout='nonlinear')
    ilqr_ss = iLQR_Tracking_SE3(dynamics, cost, N, hessians=False, rollout='nonlinear')
    xs_ms, us_ms, J_hist_ms, _, _, grad_hist_ms, defect_hist_ms = ilqr_ms.fit(
        x0, us_init, n_iterations=max_iterations, tol_grad_norm=tol, on_iteration=on_iteration_ms_se3)
    xs_ss, us_ss, J_hist_ss, _, _, grad_hist_ss = ilqr_ss.fit(
        x0, us_init, n_iterations=max_iterations, tol_grad_norm=tol, on_iteration=on_iteration_ss_se3)
    data = {
        'prob': {'J': J, 'dt': dt, 'q_ref': q_ref, 'xi_ref': xi_ref, 'x0': x0, 'Q': Q, 'P': P, 'R': R},
        'ms_se3': {'xs': xs_ms, 'us': us_ms, 'J_hist': J_hist_ms, 'grad_hist': grad_hist_ms, 'defect_hist': defect_hist_ms},
        'ss_se3': {'xs': xs_ss, 'us': us_ss, 'J_hist': J_hist_ss, 'grad_hist': grad_hist_ss},
    }
    data['ms_se3']['max_dyn_err'] = max(err_dyn(xs_ms[k], xs_ms[k + 1], dt) for k in range(N))
    if save_to:
        results_io.save_results(save_to, data)
        print("Results saved to", save_to)
    return data


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3] or ["drone"]), save_to=(sys.argv[3] if len(sys.argv) > 3 else None))
