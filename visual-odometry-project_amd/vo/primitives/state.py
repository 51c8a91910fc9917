"""Current/previous pose and frame bookkeeping (reference: src/vo/primitives/state.py)."""
import numpy as np

from vo.helpers import to_cartesian_coordinates, to_homogeneous_coordinates
from vo.primitives.frame import Frame
from vo.primitives.matches import Matches
from vo.sensors import Camera


def _as_4x4(pose: np.ndarray) -> np.ndarray:
    if pose.shape == (3, 4):
        pose = np.concatenate((pose, np.array([[0, 0, 0, 1]])), axis=0)
    return pose


class State:
    def __init__(self, initial_frame: Frame, bearing_threshold: float = 0.0075) -> None:
        self.curr_pose = np.eye(4)
        self.curr_frame = initial_frame
        self.prev_pose = None
        self.prev_frame = None
        self._bearing_threshold = bearing_threshold

    def update_from_matches(self, matches: Matches) -> None:
        self.prev_frame, self.prev_pose = self.curr_frame, self.curr_pose
        self.curr_frame, self.curr_pose = matches.frame2, None

    def update_with_local_pose(self, pose: np.ndarray) -> None:
        """`pose` maps previous-camera coordinates to current-camera ones (state.py:24-36)."""
        self.curr_pose = self.prev_pose @ np.linalg.inv(_as_4x4(pose))
        self.curr_frame.features.set_pose_for_new_tracks(self.curr_pose)

    def update_with_world_pose(self, pose: np.ndarray) -> None:
        """`pose` maps world coordinates to current-camera ones (state.py:38-50)."""
        self.curr_pose = np.linalg.inv(_as_4x4(pose))
        self.curr_frame.features.set_pose_for_new_tracks(self.curr_pose)

    def update_with_local_landmarks(self, landmarks: np.ndarray, keypoints_mask: np.ndarray) -> None:
        """Landmarks given in the previous camera's frame (state.py:52-67)."""
        world = to_cartesian_coordinates(self.prev_pose @ to_homogeneous_coordinates(landmarks))
        self.update_with_world_landmarks(world, keypoints_mask)

    def update_with_world_landmarks(self, landmarks: np.ndarray, keypoints_mask: np.ndarray) -> None:
        feats = self.curr_frame.features
        assert np.sum(keypoints_mask) == len(landmarks), "Mismatch in length"
        assert np.all(feats.state[keypoints_mask] == 1), "Already triangulated point"
        feats.landmarks[keypoints_mask] = landmarks
        feats.state[keypoints_mask] = 2
        self._check_landmarks()
        assert not np.any(np.isnan(feats.landmarks[feats.state == 2])), "NaN in triangulated landmarks"

    def _check_landmarks(self) -> None:
        """Landmarks behind the current or the previous camera are dropped (state.py:90-107)."""
        feats = self.curr_frame.features
        hom = to_homogeneous_coordinates(feats.landmarks)
        z_curr = to_cartesian_coordinates(np.linalg.inv(self.curr_pose) @ hom)[:, 2].flatten()
        z_prev = to_cartesian_coordinates(np.linalg.inv(self.prev_pose) @ hom)[:, 2].flatten()
        behind = (z_curr < 0) | (z_prev < 0)
        feats.landmarks[behind] = np.nan
        self.reset_outliers(behind)

    def get_frame(self) -> Frame:
        return self.curr_frame

    def get_pose(self) -> np.ndarray:
        return self.curr_pose

    def get_landmarks(self) -> np.ndarray:
        return self.curr_frame.features.landmarks

    def get_keypoints(self) -> np.ndarray:
        return self.curr_frame.features.keypoints

    def compute_candidates(self) -> None:
        """Matched, not yet triangulated tracks whose bearing angle is large enough (state.py:135-160)."""
        feats = self.curr_frame.features
        start_poses = feats.matched_candidate_inliers_poses
        end_poses = np.stack([self.curr_pose] * start_poses.shape[0], axis=0)
        angles = self._calculate_bearing_angle(self.curr_frame.sensor, start_poses, end_poses,
                                               feats.matched_candidate_inliers_tracks,
                                               feats.matched_candidate_inliers_keypoints)
        feats.candidate_mask[feats.matched_candidate_inliers] = angles >= self._bearing_threshold

    def reset_outliers(self, outliers: np.ndarray) -> None:
        """Outliers go back to 'unmatched' and restart their track here (state.py:162-172)."""
        feats = self.curr_frame.features
        feats.state[outliers] = 0
        feats.tracks[outliers] = feats.keypoints[outliers]
        feats.poses[outliers] = self.curr_pose

    def _calculate_bearing_angle(self, camera: Camera, T1: np.ndarray, T2: np.ndarray, points1: np.ndarray,
                                 points2: np.ndarray) -> np.ndarray:
        """Angle between the viewing rays at the start and the end of each track (state.py:174-219)."""
        assert len(points1) == len(points2), "Points must have same length"
        assert points1.ndim == 3, "Points must have three dimensions"
        assert points2.ndim == 3, "Points must have three dimensions"
        assert not np.any(np.isnan(points1)), "Points1 contains invalid points"
        assert not np.any(np.isnan(points2)), "Points2 contains invalid points"
        assert not np.any(np.isnan(T1)), "Invalid start poses"
        assert not np.any(np.isnan(T2)), "Invalid end poses"
        ray1 = np.matmul(T1[:, :3, :3], camera.to_normalized_image_coordinates(points1)).reshape(-1, 3)
        ray2 = np.matmul(T2[:, :3, :3], camera.to_normalized_image_coordinates(points2)).reshape(-1, 3)
        cosine = np.sum(ray1 * ray2, axis=-1) / (np.linalg.norm(ray1, axis=-1) * np.linalg.norm(ray2, axis=-1))
        return np.arccos(cosine)
