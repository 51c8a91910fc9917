"""Homogeneous-coordinate and SE(3) helpers (reference: src/vo/helpers.py).

Tiny host-side NumPy/SciPy utilities; they stay on the host as in the reference."""
import numpy as np
from scipy.linalg import expm, logm


def to_homogeneous_coordinates(points: np.ndarray) -> np.ndarray:
    """(N, D, 1) -> (N, D+1, 1) with a trailing 1 (helpers.py:5-15)."""
    assert points.ndim == 3, "Points must have three dimensions"
    ones = np.ones((points.shape[0], 1, 1))
    return np.concatenate((points, ones), axis=-2)


def to_cartesian_coordinates(points: np.ndarray) -> np.ndarray:
    """(N, D+1, 1) -> (N, D, 1), dividing by the last row (helpers.py:18-28)."""
    assert points.ndim == 3, "Points must have three dimensions"
    return points[:, :-1] / points[:, -1:]


def normalize_points(points: np.ndarray):
    """Hartley normalisation: zero mean, RMS distance sqrt(D) (helpers.py:31-54).
    Returns (normalised points (N, D, 1), T (D+1, D+1))."""
    dim = points.shape[1]
    centre = np.mean(points, axis=0, keepdims=True)
    rms = np.sqrt(np.mean(np.sum((points - centre) ** 2, axis=-2)))
    scale = np.sqrt(dim) / rms
    T = np.diag([scale] * dim + [1])
    T[:-1, -1:] = -scale * centre.reshape(dim, 1)
    return to_cartesian_coordinates(T @ to_homogeneous_coordinates(points)), T


def to_skew_symmetric_matrix(v: np.ndarray) -> np.ndarray:
    """[v]_x for a (3, 1) vector or a stack (N, 3, 1) (helpers.py:57-83)."""
    assert (v.ndim == 2 and v.shape == (3, 1)) or (
        v.ndim == 3 and v.shape[1:] == (3, 1)
    ), "Vector must be a single 3D vector or an array of 3D vectors"
    single = v.ndim == 2
    w = v.reshape(-1, 3)
    out = np.zeros((w.shape[0], 3, 3))
    out[:, 0, 1], out[:, 0, 2] = -w[:, 2], w[:, 1]
    out[:, 1, 0], out[:, 1, 2] = w[:, 2], -w[:, 0]
    out[:, 2, 0], out[:, 2, 1] = -w[:, 1], w[:, 0]
    return out.squeeze() if single else out


def skew_matrix_to_cross(M):
    """Inverse of to_skew_symmetric_matrix for one 3x3 matrix (helpers.py:130-142)."""
    return np.array([-M[1, 2], M[0, 2], -M[0, 1]])


def twist_to_H_matrix(twist):
    """[v; w] (6,) -> 4x4 rigid transform via the matrix exponential (helpers.py:86-102)."""
    v, w = twist[:3], twist[3:]
    se3 = np.zeros((4, 4))
    se3[:3, :3] = to_skew_symmetric_matrix(w.reshape(3, 1))
    se3[:3, 3] = v
    return expm(se3)


def H_matrix_to_twist(H):
    """4x4 rigid transform -> [v; w] via the matrix logarithm (helpers.py:105-127)."""
    se3 = logm(H)
    return np.concatenate([se3[:3, 3], skew_matrix_to_cross(se3[:3, :3])])
