"""NumPy oracle: two-view bootstrap.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates src/vo/landmarks/triangulation.py:110-350 (use_opencv=False route) and src/vo/helpers.py:31-54:
  find_fundamental_matrix        _find_fundamental_matrix (:165-222)
  find_fundamental_matrix_ransac _find_fundamental_matrix_ransac (:110-163) around oracle/ransac_np.Ransac
  decompose_essential_matrix     _decompose_essential_matrix (:245-277)
  find_relative_pose             _find_relative_pose (:279-350), its DLT passes by oracle/dlt_np
Pinned by tests/golden/bootstrap.npz (the reference's own outputs, tools/make_golden.py golden_bootstrap).
`epipolar_errors` / `find_fundamental_matrix_ransac_px` restate the variant the product runs for use_opencv=True
(cv2.findFundamentalMat(FM_RANSAC) is not reproducible: pixel threshold on the epipolar distance, per-sample
normalisation, at most 2000 iterations); that route has no reference fixture -- parity unpinned."""
import numpy as np

from oracle import dlt_np, ransac_np


def to_h(p):
    return np.concatenate([p, np.ones((p.shape[0], 1, 1))], axis=1)


def normalize_points(points):
    """helpers.py:31-54."""
    D = points.shape[1]
    mu = np.mean(points, axis=0, keepdims=True)
    sigma = np.sqrt(np.mean(np.sum((points - mu) ** 2, axis=-2)))
    s = np.sqrt(D) / sigma
    T = np.diag([s] * D + [1])
    T[:-1, -1:] = -s * mu.reshape(D, 1)
    pts = T @ to_h(points)
    return pts[:, :-1] / pts[:, -1:], T


def find_fundamental_matrix(points1, points2, is_normalized=False):
    if not is_normalized:
        points1, T1 = normalize_points(points1)
        points2, T2 = normalize_points(points2)
    p1, p2 = to_h(points1), to_h(points2)
    Q = np.empty((p1.shape[0], 9))
    for i in range(p1.shape[0]):
        Q[i] = np.kron(p1[i], p2[i]).T
    _, _, Vh = np.linalg.svd(Q, full_matrices=True)
    F = Vh[-1, :].reshape(3, 3).T
    U, S, Vh = np.linalg.svd(F)
    S[-1] = 0
    F = U @ np.diag(S) @ Vh
    return F if is_normalized else T2.T @ F @ T1


def algebraic_errors(F, points):
    p1, p2 = to_h(points[:, 0]), to_h(points[:, 1])
    return np.sum((p2.transpose((0, 2, 1)) @ F @ p1) ** 2, axis=(1, 2))


def epipolar_errors(F, points):
    """larger squared distance to the epipolar lines of the two images, elementwise (no BLAS: a fixed summation order)"""
    x1, y1, x2, y2 = points[:, 0, 0, 0], points[:, 0, 1, 0], points[:, 1, 0, 0], points[:, 1, 1, 0]
    l2x, l2y, l2z = F[0, 0] * x1 + F[0, 1] * y1 + F[0, 2], F[1, 0] * x1 + F[1, 1] * y1 + F[1, 2], F[2, 0] * x1 + F[2, 1] * y1 + F[2, 2]
    l1x, l1y = F[0, 0] * x2 + F[1, 0] * y2 + F[2, 0], F[0, 1] * x2 + F[1, 1] * y2 + F[2, 1]
    e = x2 * l2x + y2 * l2y + l2z
    num = e * e
    return np.maximum(num / (l1x * l1x + l1y * l1y), num / (l2x * l2x + l2y * l2y))


def find_fundamental_matrix_ransac(points1, points2, threshold, outlier_ratio, confidence):
    p1, T1 = normalize_points(points1)
    p2, T2 = normalize_points(points2)
    rs = ransac_np.Ransac(8, np.stack([p1, p2], axis=1), lambda pop: find_fundamental_matrix(pop[:, 0], pop[:, 1], True),
                          algebraic_errors, threshold, outlier_ratio, confidence)
    F, inl = rs.find_best_model()
    return T2.T @ F @ T1, inl, rs


def find_fundamental_matrix_ransac_px(points1, points2, threshold_px, outlier_ratio, confidence, max_iterations=2000):
    rs = ransac_np.Ransac(8, np.stack([points1, points2], axis=1),
                          lambda pop: find_fundamental_matrix(pop[:, 0], pop[:, 1], False), epipolar_errors,
                          threshold_px ** 2, outlier_ratio, confidence, max_iterations)
    F, inl = rs.find_best_model()
    return F, inl, rs


def decompose_essential_matrix(E):
    U, _, Vh = np.linalg.svd(E)
    T = U[:, 2:]
    W = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]])
    R = np.stack([U @ W @ Vh, U @ W.T @ Vh])
    for i in range(2):
        if np.linalg.det(R[i]) < 0:
            R[i] *= -1
    M = np.zeros((4, 3, 4))
    for i in range(2):
        for j in range(2):
            M[2 * i + j] = np.concatenate([R[j], (-1) ** i * T], axis=-1)
    return M


def find_relative_pose(points1, points2, K1, K2, F, inliers=None):
    E = K2.T @ F @ K1
    p1_in, p2_in = (points1, points2) if inliers is None else (points1[inliers], points2[inliers])
    M2 = decompose_essential_matrix(E)
    M1 = np.hstack((np.eye(3), np.zeros((3, 1))))
    best_valid, best_in, best_M = -1, None, None
    for m in range(4):
        X1 = dlt_np.linear_triangulation(p1_in[:, :, 0], p2_in[:, :, 0], K1 @ M1, K2 @ M2[m]).reshape(-1, 3, 1)
        X2 = M2[m][:, :3] @ X1 + M2[m][:, 3:]
        front = ((X1[:, -1] >= 0) & (X2[:, -1] >= 0)).flatten()
        if front.sum() > best_valid:
            best_valid, best_in, best_M = front.sum(), front, M2[m]
    X = dlt_np.linear_triangulation(points1[:, :, 0], points2[:, :, 0], K1 @ M1, K2 @ best_M).reshape(-1, 3, 1)
    if inliers is None:
        return best_M, X, best_in, M2
    mask = np.zeros(points1.shape[0], dtype=bool)
    mask[inliers] = best_in
    return best_M, X, mask, M2
