"""Landmark triangulation and two-view bootstrap (reference: src/vo/landmarks/triangulation.py).

Every triangulation is the batched HIP DLT kernel (vo_triangulate_dlt).  The two-view
bootstrap (8-point F, optional RANSAC, E decomposition, cheirality vote) runs once per
sequence and stays host NumPy as in the reference; its four cheirality passes use the
same DLT kernel.  `use_opencv` is accepted for signature compatibility; both values run
these routines (the reference's cv2.findFundamentalMat / cv2.triangulatePoints have no
counterpart here)."""
import numpy as np

from vo import _native
from vo.algorithms import RANSAC
from vo.helpers import normalize_points, to_homogeneous_coordinates
from vo.primitives import Features, Matches
from vo.sensors import Camera


class LandmarksTriangulator:
    def __init__(self, camera1: Camera, camera2: Camera, use_ransac: bool = True, outlier_ratio: float = 0.9,
                 ransac_threshold: float = 3.0, ransac_confidence=0.99, use_opencv: bool = True,
                 context=None) -> None:
        self.camera1 = camera1
        self.camera2 = camera2
        self._use_ransac = use_ransac
        self._outlier_ratio = outlier_ratio
        self._ransac_reproj_threshold = ransac_threshold
        self._ransac_confidence = ransac_confidence
        self._use_opencv = use_opencv
        self._ctx = context

    def _context(self):
        if self._ctx is None:
            self._ctx = _native.default_context()
        return self._ctx

    # ---- per-frame path ----
    def triangulate_candidates(self, features: Features, current_pose: np.ndarray) -> np.ndarray:
        """World landmarks of the candidate tracks from their start and end observations; one
        projection matrix per track start (triangulation.py:38-86)."""
        cand = features.candidate_mask
        start, end = features.tracks[cand], features.keypoints[cand]
        proj1 = self.camera1.intrinsic_matrix @ np.linalg.inv(features.poses[cand])[:, :3]
        proj2 = self.camera2.intrinsic_matrix @ np.linalg.inv(current_pose)[:3]
        if start.shape[0] == 0:
            return np.zeros((0, 3, 1))
        return self._context().triangulate_dlt(start, end, proj1, proj2).reshape(-1, 3, 1)

    def _linear_triangulation(self, points1, points2, C1, C2):
        """DLT of N correspondences with shared projection matrices (triangulation.py:352-389)."""
        assert points1.shape == points2.shape, "Input points dimension mismatch"
        assert points1.shape[1] == 2, "Points must have two rows for (u,v)"
        assert points1.shape[2] == 1, "Points must be a column vector"
        assert C1.shape == (3, 4) and C2.shape == (3, 4), "Matrix C1 and C2 must be 3 rows and 4 columns [R T]"
        if points1.shape[0] == 0:
            return np.zeros((0, 3, 1))
        return self._context().triangulate_dlt(points1, points2, np.asarray(C1, float), np.asarray(C2, float)).reshape(-1, 3, 1)

    # ---- bootstrap ----
    def triangulate_matches(self, matches: Matches):
        """(M (3,4), landmarks (N,3,1)[, inlier mask]) from the matched candidates of two
        frames (triangulation.py:88-108)."""
        points1 = matches.frame1.features.matched_candidate_inliers_keypoints
        points2 = matches.frame2.features.matched_candidate_inliers_keypoints
        return self._find_relative_pose(points1, points2)

    def _find_fundamental_matrix(self, points1: np.ndarray, points2: np.ndarray, is_normalized: bool = False):
        """Normalised 8-point algorithm with the rank-2 constraint (triangulation.py:165-222)."""
        assert points1.shape == points2.shape, "Input points dimension mismatch"
        assert points1.shape[0] >= 8, "Not enough points for 8-point algorithm"
        assert points1.shape[1] == 2, "Points must have two rows for (u,v)"
        assert points1.shape[2] == 1, "Points must be a column vector"
        if not is_normalized:
            points1, T1 = normalize_points(points1)
            points2, T2 = normalize_points(points2)
        p1 = to_homogeneous_coordinates(points1)[:, :, 0]
        p2 = to_homogeneous_coordinates(points2)[:, :, 0]
        Q = (p1[:, :, None] * p2[:, None, :]).reshape(-1, 9)         # rows kron(p1_i, p2_i)
        _, _, Vh = np.linalg.svd(Q, full_matrices=True)
        F = Vh[-1, :].reshape(3, 3).T
        U, S, Vh = np.linalg.svd(F)
        S[-1] = 0
        F = U @ np.diag(S) @ Vh
        return F if is_normalized else T2.T @ F @ T1

    def _find_fundamental_matrix_ransac(self, points1: np.ndarray, points2: np.ndarray):
        """8-point inside RANSAC on normalised points, algebraic error (triangulation.py:110-163)."""
        def model_fn(population):
            return self._find_fundamental_matrix(population[:, 0], population[:, 1], is_normalized=True)

        def error_fn(F, points):
            p1 = to_homogeneous_coordinates(points[:, 0])
            p2 = to_homogeneous_coordinates(points[:, 1])
            return np.sum((p2.transpose((0, 2, 1)) @ F @ p1) ** 2, axis=(1, 2))

        if self._use_opencv:
            # cv2.findFundamentalMat(FM_RANSAC, ransacReprojThreshold, confidence) as main.py:185-193
            # configures it: a correspondence is an inlier when its squared distance to the
            # epipolar line, in pixels and in both images, is within the threshold squared; F is
            # re-fitted on all inliers (triangulation.py:126-133 hands this to OpenCV).
            def model_px(population):
                return self._find_fundamental_matrix(population[:, 0], population[:, 1], is_normalized=False)

            def error_px(F, points):
                p1 = to_homogeneous_coordinates(points[:, 0])[:, :, 0]
                p2 = to_homogeneous_coordinates(points[:, 1])[:, :, 0]
                l2 = p1 @ F.T                                          # epipolar lines in image 2
                l1 = p2 @ F                                            # epipolar lines in image 1
                num = np.sum(p2 * l2, axis=1) ** 2
                d2 = num / (l2[:, 0] ** 2 + l2[:, 1] ** 2)
                d1 = num / (l1[:, 0] ** 2 + l1[:, 1] ** 2)
                return np.maximum(d1, d2)

            ransac_px = RANSAC(s_points=8, population=np.stack([points1, points2], axis=1), model_fn=model_px,
                               error_fn=error_px, inlier_threshold=self._ransac_reproj_threshold ** 2,
                               outlier_ratio=self._outlier_ratio, confidence=self._ransac_confidence,
                               max_iterations=2000)
            return ransac_px.find_best_model()
        points1, T1 = normalize_points(points1)
        points2, T2 = normalize_points(points2)
        ransac_F = RANSAC(s_points=8, population=np.stack([points1, points2], axis=1), model_fn=model_fn,
                          error_fn=error_fn, inlier_threshold=self._ransac_reproj_threshold,
                          outlier_ratio=self._outlier_ratio, confidence=self._ransac_confidence)
        F, inliers = ransac_F.find_best_model()
        return T2.T @ F @ T1, inliers

    def _find_essential_matrix(self, points1: np.ndarray, points2: np.ndarray):
        """E = K2^T F K1 (triangulation.py:224-243)."""
        K1, K2 = self.camera1.intrinsic_matrix, self.camera2.intrinsic_matrix
        if self._use_ransac:
            F, inliers = self._find_fundamental_matrix_ransac(points1, points2)
            return K2.T @ F @ K1, inliers
        return K2.T @ self._find_fundamental_matrix(points1, points2) @ K1

    def _decompose_essential_matrix(self, E: np.ndarray) -> np.ndarray:
        """The four [R | +-T] candidates (triangulation.py:245-277)."""
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

    def _find_relative_pose(self, points1: np.ndarray, points2: np.ndarray):
        """Relative pose camera1 -> camera2 by cheirality vote over the four decompositions, and
        the triangulation of ALL input points with the winner (triangulation.py:279-350)."""
        if self._use_ransac:
            E, inliers = self._find_essential_matrix(points1, points2)
            p1_in, p2_in = points1[inliers], points2[inliers]
        else:
            E = self._find_essential_matrix(points1, points2)
            p1_in, p2_in = points1, points2
        M2 = self._decompose_essential_matrix(E)
        M1 = np.hstack((np.eye(3), np.zeros((3, 1))))
        K1, K2 = self.camera1.intrinsic_matrix, self.camera2.intrinsic_matrix
        best_valid, best_inliers, best_M = -1, None, None
        for m in range(M2.shape[0]):
            X1 = self._linear_triangulation(p1_in, p2_in, K1 @ M1, K2 @ M2[m])
            X2 = M2[m][:, :3] @ X1 + M2[m][:, 3:]
            in_front = ((X1[:, -1] >= 0) & (X2[:, -1] >= 0)).flatten()
            if in_front.sum() > best_valid:
                best_valid, best_inliers, best_M = in_front.sum(), in_front, M2[m]
        landmarks = self._linear_triangulation(points1, points2, K1 @ M1, K2 @ best_M)
        if self._use_ransac:
            mask = np.zeros((points1.shape[0],), dtype=bool)
            mask[inliers] = best_inliers
            return best_M, landmarks, mask
        return best_M, landmarks
