"""Camera pose from 3D-2D correspondences: P3P inside RANSAC, then a twist least-squares
refinement (reference: src/vo/pose_estimation/p3p.py).

The reference evaluates one hypothesis per Python iteration through OpenCV
(cv2.solvePnP(P3P), cv2.projectPoints).  Here every batch of sampled hypotheses is
solved and scored by two HIP kernels (vo_p3p_hypotheses) and the sequential RANSAC rule
is replayed on the host, so inlier masks and iteration counts are those of the
sequential loop.  `use_opencv` is kept for signature compatibility: both values run
this solver; with use_opencv=True the threshold is taken in pixels (as
cv2.solvePnPRansac does, p3p.py:143-152) instead of squared pixels."""
import numpy as np

from vo import _native
from vo.algorithms import RANSAC
from vo.primitives import Features


def _project(points_3d, R, t, K):
    """cv2.projectPoints without distortion: x' = X' * (1/Z'), u = x' * fx + cx; (N, 2, 1)."""
    Xc = R[np.newaxis] @ points_3d + np.asarray(t).reshape(1, 3, 1)
    iz = 1.0 / Xc[:, 2]
    u = (Xc[:, 0] * iz) * K[0, 0] + K[0, 2]
    v = (Xc[:, 1] * iz) * K[1, 1] + K[1, 2]
    return np.stack([u, v], axis=1)


class P3PPoseEstimator:
    def __init__(self, intrinsic_matrix: np.ndarray, inlier_threshold: float, use_opencv: bool = True,
                 outlier_ratio: float = 0.9, confidence: float = 0.99, max_iterations: int = 10000,
                 nonlinear_refinement: bool = True, batch_size: int = 1000, context=None) -> None:
        self._use_opencv = use_opencv
        self.intrinsic_matrix = intrinsic_matrix
        self.inlier_threshold = inlier_threshold
        self.outlier_ratio = outlier_ratio
        self.confidence = confidence
        self.max_iterations = max_iterations
        self.nonlinear_refinement = nonlinear_refinement
        self.batch_size = batch_size
        self._ctx = context
        K = np.asarray(intrinsic_matrix, dtype=np.float64)
        thr = float(inlier_threshold) ** 2 if use_opencv else float(inlier_threshold)
        self._thr_sq = thr

        def model_fn(points: np.ndarray, K: np.ndarray = K):
            """Pose from 4 sampled correspondences, or None (p3p.py:51-79)."""
            assert points.shape[0] == 4, "P3P requires 4 point correspondences"
            X = np.stack(points[:, 0]).reshape(4, 3)
            x = np.stack(points[:, 1]).reshape(4, 2)
            R, t, valid, _ = self._context().p3p_hypotheses(X, x, K, np.arange(4, dtype=np.int32)[None], thr)
            return (R[0], t[0].reshape(3, 1)) if valid[0] else None

        def error_fn(model, population, K: np.ndarray = K) -> np.ndarray:
            """Squared reprojection error of every correspondence (p3p.py:81-108)."""
            X = np.stack(population[:, 0]).reshape(-1, 3)
            x = np.stack(population[:, 1]).reshape(-1, 2)
            _, err = self._context().reproj_inliers(X, x, K, model[0], model[1], thr, want_err=True)
            return err

        self.ransac = RANSAC(s_points=4, population=None, model_fn=model_fn, error_fn=error_fn,
                             inlier_threshold=thr, outlier_ratio=self.outlier_ratio, confidence=self.confidence,
                             max_iterations=self.max_iterations, p3p=True)

    def _context(self):
        if self._ctx is None:
            self._ctx = _native.default_context()
        return self._ctx

    def estimate_pose(self, features: Features):
        """((R (3,3), t (3,1)), inlier mask (N,)) -- world -> camera (p3p.py:123-186)."""
        points_3d, points_2d = features.landmarks, features.keypoints
        assert points_3d.shape[1] == 3 and points_2d.shape[1] == 2, "Invalid shape."
        assert points_3d is not None and points_2d is not None, "3D landmarks and 2D keypoints must be provided."
        ctx = self._context()
        K = np.asarray(self.intrinsic_matrix, dtype=np.float64)
        X = np.ascontiguousarray(points_3d, dtype=np.float64).reshape(-1, 3)
        x = np.ascontiguousarray(points_2d, dtype=np.float64).reshape(-1, 2)
        N = X.shape[0]
        assert N >= 4, "P3P requires 4 point correspondences"

        def batch(samples):
            R, t, valid, counts, masks = ctx.p3p_hypotheses(X, x, K, samples, self._thr_sq, want_masks=True)
            return valid, counts, lambda b: ((R[b].copy(), t[b].reshape(3, 1).copy()), masks[b].copy())

        best_model, best_inlier, _ = self.ransac.find_best_model_batched(N, batch, self.batch_size)
        if self.nonlinear_refinement:
            best_model = self._nonlinear_refinement(points_3d[best_inlier], points_2d[best_inlier], best_model)
        return best_model, best_inlier

    def _nonlinear_refinement(self, points_3d, points_2d, best_model):
        """The pose that minimises the summed squared reprojection distance of the given points,
        from best_model (p3p.py:188-213).  The reference asks scipy.optimize.least_squares for it
        (twist parametrisation, numerical Jacobian, tolerances 1e-8, so it stops within ~1e-4 of
        the minimiser); the device kernel runs Gauss-Newton with the analytic Jacobian to
        convergence (vo_refine_pose)."""
        K = np.asarray(self.intrinsic_matrix, dtype=np.float64)
        R, t, _, _ = self._context().refine_pose(points_3d, points_2d, K, best_model[0], np.asarray(best_model[1]).reshape(3))
        return R, t.reshape(3, 1)
