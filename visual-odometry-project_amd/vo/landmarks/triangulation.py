"""Landmark triangulation and two-view bootstrap (reference: src/vo/landmarks/triangulation.py).

Every triangulation is the batched HIP DLT kernel (vo_triangulate_dlt).  The two-view
bootstrap runs on the device as well (csrc/bootstrap.hip): 8-point hypotheses of a whole
batch of RANSAC samples and their inlier counts in two launches, the closing fit over all
inliers, and E decomposition + the four cheirality votes + the final triangulation in one
kernel; the host keeps the reference's sequential accept / adapt rule over the counts and
its generator.  `use_opencv` is accepted for signature compatibility; both values run
these routines (the reference's cv2.findFundamentalMat / cv2.triangulatePoints have no
counterpart here)."""
import numpy as np

from vo import _native
from vo.algorithms import RANSAC
from vo.helpers import normalize_points
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
        """Normalised 8-point algorithm with the rank-2 constraint (triangulation.py:165-222): Hartley normalisation,
        normal matrix, its smallest eigenvector and the rank-2 projection in one kernel (vo_fundamental_fit)."""
        assert points1.shape == points2.shape, "Input points dimension mismatch"
        assert points1.shape[0] >= 8, "Not enough points for 8-point algorithm"
        assert points1.shape[1] == 2, "Points must have two rows for (u,v)"
        assert points1.shape[2] == 1, "Points must be a column vector"
        return self._context().fundamental_fit(points1, points2, None, normalize=not is_normalized)

    def _ransac_fundamental(self, points1, points2, threshold, normalize_samples, error_kind, max_iterations=np.inf,
                            batch_size=2048):
        """The reference's RANSAC loop (src/vo/algorithms/ransac.py:69-129, s = 8) with its model_fn / error_fn calls
        batched: every batch of samples drawn ahead from a copy of the generator is solved and scored by two launches
        (vo_fundamental_hypotheses), the sequential accept / adapt rule is replayed over the counts, the generator moves by
        what the loop consumed; the closing fit over all inliers is vo_fundamental_fit.  Returns (F, inlier mask, RANSAC)."""
        ctx = self._context()
        n = points1.shape[0]
        ransac = RANSAC(s_points=8, population=np.arange(n), model_fn=None, error_fn=None, inlier_threshold=threshold,
                        outlier_ratio=self._outlier_ratio, confidence=self._ransac_confidence, max_iterations=max_iterations)

        def batch_fn(samples):
            F, counts, masks = ctx.fundamental_hypotheses(points1, points2, samples, threshold, normalize_samples, error_kind,
                                                          want_masks="packed")
            return np.ones(len(samples), np.uint8), counts, lambda b: (F[b], ctx.unpack_mask(masks[b], n))

        _, inliers, _ = ransac.find_best_model_batched(n, batch_fn, batch_size=batch_size)
        F = ctx.fundamental_fit(points1, points2, inliers, normalize=normalize_samples)   # ransac.py:123-127
        return F, inliers, ransac

    def _find_fundamental_matrix_ransac(self, points1: np.ndarray, points2: np.ndarray):
        """8-point inside RANSAC (triangulation.py:110-163)."""
        if self._use_opencv:
            # cv2.findFundamentalMat(FM_RANSAC, ransacReprojThreshold, confidence) as main.py:185-193 configures it is
            # not reproducible (OpenCV's own sampler).  This route: a correspondence is an inlier when its squared
            # distance to the epipolar line, in pixels and in both images, is within the threshold squared; every sample
            # is fitted in its own Hartley frame; at most 2000 iterations; F re-fitted on all inliers.
            F, inliers, _ = self._ransac_fundamental(points1, points2, self._ransac_reproj_threshold ** 2, True, 1,
                                                     max_iterations=2000)
            return F, inliers
        # the reference's own route: the whole population normalised once (helpers.py:31-54), algebraic error
        p1, T1 = normalize_points(points1)
        p2, T2 = normalize_points(points2)
        F, inliers, _ = self._ransac_fundamental(p1, p2, self._ransac_reproj_threshold, False, 0)
        return T2.T @ F @ T1, inliers

    def _find_essential_matrix(self, points1: np.ndarray, points2: np.ndarray):
        """E = K2^T F K1 (triangulation.py:224-243)."""
        K1, K2 = self.camera1.intrinsic_matrix, self.camera2.intrinsic_matrix
        if self._use_ransac:
            F, inliers = self._find_fundamental_matrix_ransac(points1, points2)
            return K2.T @ F @ K1, inliers
        return K2.T @ self._find_fundamental_matrix(points1, points2) @ K1

    def _decompose_essential_matrix(self, E: np.ndarray) -> np.ndarray:
        """The four [R | +-T] candidates (triangulation.py:245-277); the same set as the reference's, in the order this
        library's SVD gives (vo_essential_decompose)."""
        return self._context().essential_decompose(E)

    def _find_relative_pose(self, points1: np.ndarray, points2: np.ndarray):
        """Relative pose camera1 -> camera2 by cheirality vote over the four decompositions, and
        the triangulation of ALL input points with the winner (triangulation.py:279-350): one kernel
        (vo_relative_pose) behind the fundamental matrix."""
        K1 = np.asarray(self.camera1.intrinsic_matrix, np.float64)
        K2 = np.asarray(self.camera2.intrinsic_matrix, np.float64)
        if self._use_ransac:
            F, inliers = self._find_fundamental_matrix_ransac(points1, points2)
        else:
            F, inliers = self._find_fundamental_matrix(points1, points2), None
        M, X, mask, _ = self._context().relative_pose(points1, points2, K1, K2, F, inliers)
        if self._use_ransac:
            return M, X.reshape(-1, 3, 1), mask
        return M, X.reshape(-1, 3, 1)
