"""NumPy oracle: DLT triangulation.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates LandmarksTriangulator._linear_triangulation / triangulate_candidates
(reference src/vo/landmarks/triangulation.py:352-389, 38-86; skew matrix from
src/vo/helpers.py:57-83).  Pinned to 1e-9 by tests/golden/dlt_cameras.npz.
"""
import numpy as np


def _skew_rows(x):
    """[x]_x for x = (u, v, 1): (n, 3, 3)."""
    n = x.shape[0]
    u, v = x[:, 0], x[:, 1]
    S = np.zeros((n, 3, 3))
    S[:, 0, 1] = -1.0
    S[:, 0, 2] = v
    S[:, 1, 0] = 1.0
    S[:, 1, 2] = -u
    S[:, 2, 0] = -v
    S[:, 2, 1] = u
    return S


def linear_triangulation(x1, x2, C1, C2):
    """x1, x2: (n, 2); C1: (3, 4) or (n, 3, 4); C2: (3, 4).  Returns (n, 3).

    Per point: A = [[x1]_x C1; [x2]_x C2] (6x4), SVD, last right-singular
    vector, de-homogenise (triangulation.py:379-389).  np.linalg.svd on a stack
    runs the same LAPACK gesdd per matrix as the reference's per-point call."""
    x1 = np.asarray(x1, np.float64).reshape(-1, 2)
    x2 = np.asarray(x2, np.float64).reshape(-1, 2)
    n = x1.shape[0]
    C1 = np.asarray(C1, np.float64)
    C1 = np.broadcast_to(C1, (n, 3, 4)) if C1.ndim == 2 else C1
    A = np.concatenate([_skew_rows(x1) @ C1, _skew_rows(x2) @ np.asarray(C2, np.float64)[None]], axis=1)
    _, _, Vh = np.linalg.svd(A, full_matrices=False)
    P = Vh[:, -1, :]
    return P[:, :3] / P[:, 3:]


def candidate_projections(K, poses_start, current_pose):
    """proj1[i] = K @ inv(pose_start_i)[:3], proj2 = K @ inv(current_pose)[:3]
    (triangulation.py:53-57)."""
    ext1 = np.linalg.inv(poses_start)[:, :3]
    ext2 = np.linalg.inv(current_pose)[:3]
    return K @ ext1, K @ ext2
