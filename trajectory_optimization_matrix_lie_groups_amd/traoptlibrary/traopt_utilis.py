"""Lie-algebra bookkeeping helpers with the reference's names and conventions
(reference traoptlibrary/traopt_utilis.py:13-399): twist order [omega, v]; manif order [v, omega].
Pure index shuffling and scipy conversions -- host code in the reference too."""
import numpy as np
from scipy.spatial.transform import Rotation


def skew(w):
    """traopt_utilis.py:13-24"""
    w = np.asarray(w)
    if w.shape == (3,) or w.shape == (3, 1):
        w = w.reshape(-1)
        return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    raise ValueError("Input must be a 3d np or jnp vector")


def unskew(omega_hat):
    """traopt_utilis.py:26-41"""
    omega_hat = np.asarray(omega_hat)
    if omega_hat.shape == (3, 3):
        return np.array([omega_hat[2, 1], omega_hat[0, 2], omega_hat[1, 0]])
    raise ValueError("Input must be a 3x3 np or jnp matrix")


def se3_hat(xi):
    """traopt_utilis.py:43-55"""
    xi = np.asarray(xi)
    if xi.shape == (6,) or xi.shape == (6, 1):
        xi = xi.reshape(-1)
        return np.block([[skew(xi[:3]), xi[3:6].reshape(3, 1)], [np.zeros((1, 3)), 0]])
    raise ValueError("Input must be a 6d np or jnp array")


def se3_vee(se3_mat):
    """traopt_utilis.py:57-73"""
    se3_mat = np.asarray(se3_mat)
    if se3_mat.shape == (4, 4):
        return np.concatenate((unskew(se3_mat[:3, :3]), se3_mat[:3, 3].reshape(3,)))
    raise ValueError("Input must be a 4x4 np or jnp array representing an se(3) matrix")


def adjoint(xi):
    """traopt_utilis.py:75-88"""
    xi = np.asarray(xi)
    if xi.shape == (6,) or xi.shape == (6, 1):
        xi = xi.reshape(6,)
        return np.block([[skew(xi[:3]), np.zeros((3, 3))], [skew(xi[3:]), skew(xi[:3])]])
    raise ValueError("Input must be a 6-d np or jnp vector")


def coadjoint(xi):
    """traopt_utilis.py:90-92"""
    return adjoint(xi).T


def quat2rotm(quat):
    """traopt_utilis.py:159-161 (scalar-first quaternion)"""
    return Rotation.from_quat(quat, scalar_first=True).as_matrix()


def quat2euler(quat):
    return Rotation.from_quat(quat, scalar_first=True).as_euler("zxy", degrees=True)


def rotm2quat(m):
    """traopt_utilis.py:167-181 (returns scalar-first)"""
    q1, q2, q3, q0 = Rotation.from_matrix(m).as_quat()
    return np.array([q0, q1, q2, q3])


def rotm2euler(m, order="zxy"):
    return np.array(Rotation.from_matrix(m).as_euler(order or "zxy", degrees=True))


def SE32absangle(m):
    if m.shape != (4, 4):
        raise ValueError("The input must be a 4x4 SE3 matrix")
    return np.rad2deg(np.arccos((np.trace(m[:3, :3]) - 1) / 2))


def rotm2absangle(m):
    if m.shape != (3, 3):
        raise ValueError("The input must be a 3x3 rotation matrix")
    return np.rad2deg(np.arccos((np.trace(m) - 1) / 2))


def parallel_SE32absangle(m_list):
    return np.array([SE32absangle(m) for m in m_list])


def parallel_rotm2absangle(m_list):
    return np.array([rotm2absangle(m) for m in m_list])


def parallel_rotm2euler(m_list, order):
    return np.array([rotm2euler(m, order) for m in m_list])


def quatpos2SE3(x7):
    """traopt_utilis.py:252-271"""
    x7 = np.asarray(x7)
    if x7.shape == (7,) or x7.shape == (7, 1):
        x7 = x7.reshape(7)
        return np.block([[quat2rotm(x7[:4]), x7[4:].reshape(3, 1)], [np.zeros((1, 3)), 1]])
    raise ValueError("Input must be a 7-d np or jnp vector")


def rotmpos2SE3(m, x):
    """traopt_utilis.py:273-289"""
    x = np.asarray(x)
    if x.shape == (3,) or x.shape == (3, 1):
        return np.block([[m, x.reshape(3, 1)], [np.zeros((1, 3)), 1]])
    raise ValueError("Input dimension incorrect")


def SE32quatpos(m):
    """traopt_utilis.py:299-316"""
    m = np.asarray(m)
    if m.shape == (4, 4):
        return np.concatenate((rotm2quat(m[:3, :3]), m[:3, 3])).reshape((7, 1))
    raise ValueError("Input must be a 4*4 np or jnp vector")


def is_pos_def(A):
    """traopt_utilis.py:320-329"""
    if np.array_equal(A, A.T):
        try:
            np.linalg.cholesky(A)
            return True
        except np.linalg.LinAlgError:
            return False
    return False


def Jmnf2J(J):
    """traopt_utilis.py:387-399: manif [v, w] block order -> [w, v]"""
    return np.block([[J[3:, 3:], J[3:, :3]], [J[:3, 3:], J[:3, :3]]])


def se32manifse3_coeffs(x):
    """twist [w, v] -> manif coefficient order [v, w] (traopt_utilis.py:356-367)"""
    x = np.asarray(x)
    return np.concatenate((x[3:], x[:3]))


def manifse32se3(x):
    """manif coefficient order [v, w] -> twist [w, v] (traopt_utilis.py:369-383)"""
    x = np.asarray(x.coeffs() if hasattr(x, "coeffs") else x)
    return np.concatenate((x[3:], x[:3]))


from ..manifpy_compat import SE3, SE3Tangent, SO3, SO3Tangent  # noqa: E402,F401  (the reference imports these from manifpy)


def se32manifse3(x):
    """twist [w, v] -> manif SE3Tangent ([v, w]) (traopt_utilis.py:356-367)"""
    return SE3Tangent(se32manifse3_coeffs(x))


def SE32manifSE3(x):
    """traopt_utilis.py:331-342"""
    qp = SE32quatpos(x).reshape(7)
    return SE3(position=qp[4:], quaternion=np.array([qp[1], qp[2], qp[3], qp[0]]))


def manifSE32SE3(x):
    """traopt_utilis.py:344-354"""
    return x.transform()


def parallel_SE32manifSE3(q_ref):
    return [SE32manifSE3(q) for q in q_ref]


def SO32manifSO3(x):
    """traopt_utilis.py:291-297"""
    return SO3(Rotation.from_matrix(x).as_quat())
