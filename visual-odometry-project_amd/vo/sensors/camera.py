"""Pinhole camera (reference: src/vo/sensors/camera.py)."""
import numpy as np

from vo.helpers import to_cartesian_coordinates, to_homogeneous_coordinates


class Camera:
    """Intrinsics K plus an optional world->camera pose (R, t)."""

    def __init__(self, intrinsic_matrix: np.ndarray, distortion_coeffs: np.ndarray = None,
                 R: np.ndarray = None, t: np.ndarray = None):
        self.intrinsic_matrix = intrinsic_matrix
        self.distortion_coeffs = distortion_coeffs
        self.R = R
        self.t = t

    def _require_pose(self):
        assert self.R is not None and self.t is not None, "Camera pose not set"

    @property
    def projection_matrix(self) -> np.ndarray:
        """K [R | t] (camera.py:30-36)."""
        self._require_pose()
        return self.intrinsic_matrix @ np.hstack((self.R, self.t))

    @property
    def c_T_w(self) -> np.ndarray:
        """4x4 world->camera transform (camera.py:94-100)."""
        self._require_pose()
        return np.vstack((np.hstack((self.R, self.t)), [0, 0, 0, 1]))

    def distort_points(self, points: np.ndarray) -> np.ndarray:      # camera.py:38-45 (unimplemented there too)
        pass

    def undistort(self, image: np.ndarray) -> np.ndarray:            # camera.py:47-54
        pass

    def project_points_world_frame(self, points_3d: np.ndarray) -> np.ndarray:
        """(N, 3, 1) world points -> (N, 2, 1) pixels (camera.py:56-65)."""
        self._require_pose()
        return self.project_points_camera_frame(self.R[np.newaxis] @ points_3d + self.t)

    def project_points_camera_frame(self, points_3d: np.ndarray) -> np.ndarray:
        """(N, 3, 1) camera-frame points -> (N, 2, 1) pixels (camera.py:67-78)."""
        return to_cartesian_coordinates(self.intrinsic_matrix[np.newaxis] @ points_3d)

    def to_normalized_image_coordinates(self, points_2d: np.ndarray) -> np.ndarray:
        """(N, 2, 1) pixels -> (N, 3, 1) bearing vectors with unit z (camera.py:80-92)."""
        rays = np.linalg.inv(self.intrinsic_matrix) @ to_homogeneous_coordinates(points_2d)
        assert np.allclose(rays[:, -1], 1), "Normalization not successful"
        return rays
