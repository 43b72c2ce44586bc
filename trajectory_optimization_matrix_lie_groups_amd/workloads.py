"""Synthetic workloads of the BASELINE.json configurations.

The reference paths (q_ref, xi_ref, dt) are the reference's own data files re-saved as .npz under
``data/`` (tests/golden/make_golden.py); everything else -- weights, nominal initial state, the
seeded perturbation of the batch -- follows the reference's benchmark scripts:
benchmark_SE3_tracking.py:67-79,175-190 and benchmark_drone_racing_tracking.py:56-66,168-210.
Pure NumPy host code (no solver arithmetic here).
"""
import os

import numpy as np

from .solver import TrackingProblem

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
SEED = 24234156  # the reference's seed constant (main_SE3ddp_tracking_exact.py:22)


def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def _so3_exp(w):
    th = np.linalg.norm(w)
    W = _skew(w)
    if th < 1e-8:
        return np.eye(3) + W + 0.5 * W @ W
    return np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W


def _se3_exp(tau):
    w, v = tau[:3], tau[3:]
    th = np.linalg.norm(w)
    W = _skew(w)
    if th < 1e-8:
        V = np.eye(3) + 0.5 * W
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * W @ W
    T = np.eye(4)
    T[:3, :3] = _so3_exp(w)
    T[:3, 3] = V @ v
    return T


def _rot_zxy(z, x, y):
    """scipy Rotation.from_euler('zxy', [z, x, y], degrees=True).as_matrix(): lower-case axes are
    extrinsic rotations about the fixed z, then x, then y axes."""
    z, x, y = np.deg2rad([z, x, y])
    Rz = np.array([[np.cos(z), -np.sin(z), 0], [np.sin(z), np.cos(z), 0], [0, 0, 1.0]])
    Rx = np.array([[1.0, 0, 0], [0, np.cos(x), -np.sin(x)], [0, np.sin(x), np.cos(x)]])
    Ry = np.array([[np.cos(y), 0, np.sin(y)], [0, 1.0, 0], [-np.sin(y), 0, np.cos(y)]])
    return Ry @ Rx @ Rz


def load_reference(name):
    d = np.load(os.path.join(_DATA, "ref_%s.npz" % name))
    return d["q_ref"], d["xi_ref"], float(d["dt"])


def inertia():
    return np.diag([0.5, 0.7, 0.9, 1.0, 1.0, 1.0])


def perturbed_batch(q0, xi0, B, scale_pose, scale_twist, seed=SEED):
    """q0_b = q0 Exp(delta_b), xi0_b = xi0 + eta_b."""
    rng = np.random.default_rng(seed)
    x0_q = np.empty((B, 4, 4))
    x0_xi = np.empty((B, 6))
    for b in range(B):
        delta = rng.uniform(-1, 1, 6) * scale_pose
        eta = rng.uniform(-1, 1, 6) * scale_twist
        x0_q[b] = q0 @ _se3_exp(delta) if b > 0 else q0  # member 0 is the nominal problem
        x0_xi[b] = xi0 + (eta if b > 0 else 0)
    return x0_q, x0_xi


def _extend_reference(q_ref, xi_ref, dt, N):
    """Horizons beyond the stored 201 knots of path_se3_generate_sine_2 (the reference's longer problems --
    path_se3_spiral_static_velocity N = 400, the HEAD benchmark problem N = 955, benchmark_SE3_tracking.py:49-58 --
    are not among the stored data files): the path is continued the way the reference generates its own,
    q_{i+1} = q_i Exp(xi_i dt) with a smooth twist profile (main_SE3ddp_tracking_exact_ms.py:52-85), starting from
    the last stored knot.  Synthetic data of the reference's shape, not the reference's path."""
    n0 = q_ref.shape[0]
    q = np.empty((N + 1, 4, 4)); xi = np.empty((N + 1, 6))
    q[:n0] = q_ref; xi[:n0] = xi_ref
    for i in range(n0 - 1, N):
        s = float(i - (n0 - 1))
        xi[i + 1] = xi_ref[-1] + np.array([0.4 * np.sin(s / 40.0), 0.2 * np.sin(s / 55.0), 0.3 * np.sin(s / 70.0),
                                           0.5 * np.sin(s / 45.0), 0.4 * np.sin(s / 60.0), 0.3 * np.sin(s / 80.0)])
        q[i + 1] = q[i] @ _se3_exp(xi[i] * dt)
    return q, xi


def se3_tracking(B, N=200, R_scale=1e-5, seed=SEED):
    """BASELINE metric / config 3: SE3 exact tracking, N=200, dt=0.05 on path_se3_generate_sine_2."""
    q_ref, xi_ref, dt = load_reference("se3_sine2_n200")
    if N + 1 > q_ref.shape[0]:
        q_ref, xi_ref = _extend_reference(q_ref, xi_ref, dt, N)
    q_ref, xi_ref = q_ref[: N + 1], xi_ref[: N + 1]
    Q = np.diag([25.0, 25, 25, 10, 10, 10, 1, 1, 1, 1, 1, 1])
    prob = TrackingProblem("se3", inertia(), dt, Q, np.eye(6) * R_scale, 1.5 * Q, q_ref, xi_ref)
    q0 = np.eye(4)
    q0[:3, :3] = _rot_zxy(90.0, 10.0, 45.0)
    q0[:3, 3] = q_ref[0][:3, 3] - 1.0
    xi0 = np.ones(6) * 0.1
    x0_q, x0_xi = perturbed_batch(q0, xi0, B, np.array([0.3, 0.3, 0.3, 0.5, 0.5, 0.5]), 0.1, seed)
    return prob, x0_q, x0_xi, np.zeros((B, N, 6))


def drone_tracking(B, N=400, R_scale=1e-5, seed=SEED, perturb=1.0):
    """Config 5: DroneDynamics on the first N+1 knots of path_dense_random_columns_4obj (dt=0.004).
    `perturb` scales the spread of the initial states around the nominal one."""
    q_ref, xi_ref, dt = load_reference("drone_columns_n400")
    if not 1 <= N <= q_ref.shape[0] - 1:
        raise ValueError("path_dense_random_columns_4obj (stored part) has %d knots: horizon N must be in [1, %d]"
                         % (q_ref.shape[0], q_ref.shape[0] - 1))
    q_ref, xi_ref = q_ref[: N + 1], xi_ref[: N + 1]
    Q = np.diag([25.0, 25, 25, 10, 10, 10, 1, 1, 1, 1, 1, 1])
    prob = TrackingProblem("drone", inertia(), dt, Q, np.eye(4) * R_scale, 1.5 * Q, q_ref, xi_ref)
    q0 = np.eye(4)
    q0[:3, :3] = _rot_zxy(1e-4, 0.0, 0.0)
    q0[:3, 3] = q_ref[0][:3, 3] - 0.1
    xi0 = np.ones(6) * 1e-3
    x0_q, x0_xi = perturbed_batch(q0, xi0, B, perturb * 0.1 * np.array([0.3, 0.3, 0.3, 0.5, 0.5, 0.5]), perturb * 0.01, seed)
    return prob, x0_q, x0_xi, np.zeros((B, N, 4))


def pendulum_swingup(B, xi0_scale=5.0, seed=SEED):
    """Pendulum3dDyanmics swing-up tracking (main_pendulum3d_ddp_tracking_exact_ms.py:40-122,
    benchmark_pendulum_swingup.py:50-72): path_3dpendulum_swingup (N=80, dt=0.025), J=diag(.5,.7,.9), m=1,
    length=.5, Q=diag(10,10,10,1,1,1), P=10Q, R=1e-2 I3, q0 = from_euler('xy',[10,45] deg),
    xi0 = (1,1,0)*xi0_scale (5 in the main script, 1 in the benchmark)."""
    from .solver import embed_pendulum3d
    R_ref, w_ref, dt = load_reference("pendulum_swingup_n80")
    Q6 = np.diag([10.0, 10, 10, 1, 1, 1])
    prob = embed_pendulum3d(np.diag([0.5, 0.7, 0.9]), 1.0, 0.5, dt, Q6, np.eye(3) * 1e-2, 10 * Q6, R_ref, w_ref)
    x, y = np.deg2rad([10.0, 45.0])
    Rx = np.array([[1.0, 0, 0], [0, np.cos(x), -np.sin(x)], [0, np.sin(x), np.cos(x)]])
    Ry = np.array([[np.cos(y), 0, np.sin(y)], [0, 1.0, 0], [-np.sin(y), 0, np.cos(y)]])
    q0 = np.eye(4)
    q0[:3, :3] = Ry @ Rx  # extrinsic x then y
    xi0 = np.array([1.0, 1.0, 0.0, 0, 0, 0]) * xi0_scale
    x0_q, x0_xi = perturbed_batch(q0, xi0, B, np.array([0.3, 0.3, 0.3, 0, 0, 0]), np.array([0.1, 0.1, 0.1, 0, 0, 0]), seed)
    return prob, x0_q, x0_xi, np.zeros((B, prob.N, 6))


def so3_tracking(B=1, N=100, seed=SEED):
    """Config 2: SO3 exact tracking (main_SO3ddp_tracking_exact.py:75-125): first N+1 knots of
    path_3dpendulum_8shape (dt=.04), J=diag(.5,.7,.9), Q=diag(10,10,10,1,1,1), P=10Q, R=1e-5 I3,
    x0 = (q_ref[0], xi_ref[0]); members b > 0 are perturbed in rotation / angular velocity."""
    from .solver import embed_so3
    R_ref, w_ref, dt = load_reference("so3_8shape_n249")
    if not 1 <= N <= R_ref.shape[0] - 1:
        raise ValueError("path_3dpendulum_8shape has %d knots: horizon N must be in [1, %d]"
                         % (R_ref.shape[0], R_ref.shape[0] - 1))
    R_ref, w_ref = R_ref[: N + 1], w_ref[: N + 1]
    Q6 = np.diag([10.0, 10, 10, 1, 1, 1])
    prob = embed_so3(np.diag([0.5, 0.7, 0.9]), dt, Q6, np.eye(3) * 1e-5, 10 * Q6, R_ref, w_ref)
    q0 = np.eye(4)
    q0[:3, :3] = R_ref[0]
    xi0 = np.r_[w_ref[0], 0, 0, 0]
    x0_q, x0_xi = perturbed_batch(q0, xi0, B, np.array([0.3, 0.3, 0.3, 0, 0, 0]), np.array([0.1, 0.1, 0.1, 0, 0, 0]), seed)
    return prob, x0_q, x0_xi, np.zeros((B, N, 6))


def al_tracking(B, N=200, seed=SEED):
    """Config 4: SE3 AL-DDP multiple shooting with input box constraints
    (main_SE3ddp_tracking_exact_al_ms.py:47-152): constant-twist reference xi_ref=[0,0,1,2,0,.2], dt=.01,
    Q=diag(10,10,10,1,...,1), P=10Q, R=0, InputConstraint(-10,10), nominal x0=[I,(-1,-1,-.2)],
    xi0=[0,0,.1,2,0,.2].  Returns (prob, x0_q, x0_xi, us_init, lb, ub)."""
    dt = 0.01
    xi_c = np.array([0.0, 0.0, 1.0, 2.0, 0.0, 0.2])
    step = _se3_exp(xi_c * dt)
    q_ref = np.empty((N + 1, 4, 4))
    q_ref[0] = np.eye(4)
    for i in range(N):
        q_ref[i + 1] = q_ref[i] @ step
    xi_ref = np.repeat(xi_c[None], N + 1, 0)
    Q = np.diag([10.0, 10, 10, 1, 1, 1, 1, 1, 1, 1, 1, 1])
    prob = TrackingProblem("se3", inertia(), dt, Q, np.zeros((6, 6)), 10 * Q, q_ref, xi_ref)
    q0 = np.eye(4)
    q0[:3, 3] = [-1.0, -1.0, -0.2]
    xi0 = np.array([0.0, 0.0, 0.1, 2.0, 0.0, 0.2])
    x0_q, x0_xi = perturbed_batch(q0, xi0, B, 0.3 * np.array([0.3, 0.3, 0.3, 0.5, 0.5, 0.5]), 0.05, seed)
    return prob, x0_q, x0_xi, np.zeros((B, N, 6)), -10.0 * np.ones(6), 10.0 * np.ones(6)
