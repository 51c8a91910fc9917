"""Per-frame keypoint container (reference: src/vo/primitives/features.py).

State codes: 0 unmatched, 1 matched (candidate), 2 triangulated.  All per-keypoint
arrays share their first dimension; `mask` filters every one of them except the
descriptors (as the reference does, features.py:239-265)."""
import numpy as np

_PER_POINT = ("_keypoints", "_state", "_landmarks", "_uids", "_tracks", "_poses", "_candidate_mask")


def _same_length(name, value, n):
    assert value is None or value.shape[0] == n, "Unequal number of %s and keypoints." % name


class Features:
    def __init__(self, keypoints: np.ndarray, landmarks: np.ndarray = None, uids: np.ndarray = None) -> None:
        assert keypoints.ndim == 3 and keypoints.shape[1:] == (2, 1), "Invalid shape for keypoints"
        n = keypoints.shape[0]
        self._keypoints = keypoints
        self.descriptors = None
        if landmarks is not None:
            assert landmarks.ndim == 3 and landmarks.shape[1:] == (3, 1), "Invalid shape for landmarks"
        self.landmarks = landmarks if landmarks is not None else np.full((n, 3, 1), np.nan)
        self.state = np.zeros((n,))
        self._uids = uids
        self._tracks = keypoints.copy()                                  # a new track starts at its keypoint
        self._poses = np.stack([np.eye(4)] * n) if n > 0 else np.empty((0, 4, 4))
        self._candidate_mask = np.zeros((n,), dtype=bool)

    # ---- selections by state (features.py:56-102) ----
    @property
    def matched_candidate_inliers(self) -> np.ndarray:
        return self.state == 1

    @property
    def match_inliers(self) -> np.ndarray:
        return self.state >= 1

    @property
    def triangulate_inliers(self) -> np.ndarray:
        return self.state >= 2

    @property
    def p3p_inliers(self) -> np.ndarray:
        return self.state >= 2

    @property
    def matched_candidate_inliers_tracks(self) -> np.ndarray:
        return self._tracks[self.matched_candidate_inliers]

    @property
    def matched_candidate_inliers_poses(self) -> np.ndarray:
        return self._poses[self.matched_candidate_inliers]

    @property
    def matched_candidate_inliers_keypoints(self) -> np.ndarray:
        return self._keypoints[self.matched_candidate_inliers]

    @property
    def candidate_inliers_keypoints(self) -> np.ndarray:
        return self._keypoints[self.candidate_mask]

    @property
    def matched_inliers_keypoints(self) -> np.ndarray:
        return self._keypoints[self.match_inliers]

    @property
    def triangulated_inliers_keypoints(self) -> np.ndarray:
        return self._keypoints[self.triangulate_inliers]

    @property
    def triangulated_inliers_landmarks(self) -> np.ndarray:
        return self._landmarks[self.triangulate_inliers]

    @property
    def p3p_inliers_keypoints(self) -> np.ndarray:
        return self._keypoints[self.p3p_inliers]

    # ---- per-point arrays with length checks on assignment (features.py:104-212) ----
    @property
    def keypoints(self) -> np.ndarray:
        return self._keypoints

    @keypoints.setter
    def keypoints(self, value: np.ndarray) -> None:
        assert value.shape[0] == self._keypoints.shape[0], "Unequal number of keypoints."
        self._keypoints = value

    @property
    def state(self) -> np.ndarray:
        return self._state

    @state.setter
    def state(self, value: np.ndarray) -> None:
        assert value.shape[0] == self._keypoints.shape[0], "Unequal number of state and keypoints."
        self._state = value

    @property
    def descriptors(self) -> np.ndarray:
        return self._descriptors

    @descriptors.setter
    def descriptors(self, value: np.ndarray) -> None:
        _same_length("descriptors", value, self._keypoints.shape[0])
        self._descriptors = value

    @property
    def landmarks(self) -> np.ndarray:
        return self._landmarks

    @landmarks.setter
    def landmarks(self, value: np.ndarray) -> None:
        assert value.shape[0] == self._keypoints.shape[0], "Unequal number of landmarks and keypoints."
        self._landmarks = value

    @property
    def uids(self) -> np.ndarray:
        return self._uids

    @uids.setter
    def uids(self, value: np.ndarray) -> None:
        _same_length("uids", value, self._keypoints.shape[0])
        self._uids = value

    @property
    def tracks(self) -> np.ndarray:
        return self._tracks

    @tracks.setter
    def tracks(self, value: np.ndarray) -> None:
        _same_length("tracks", value, self._keypoints.shape[0])
        self._tracks = value

    @property
    def poses(self) -> np.ndarray:
        return self._poses

    @poses.setter
    def poses(self, value: np.ndarray) -> None:
        _same_length("poses", value, self._keypoints.shape[0])
        self._poses = value

    @property
    def candidate_mask(self) -> np.ndarray:
        return self._candidate_mask

    @candidate_mask.setter
    def candidate_mask(self, value: np.ndarray) -> None:
        _same_length("candidate_mask", value, self._keypoints.shape[0])
        self._candidate_mask = value

    @property
    def length(self) -> int:
        n = self._keypoints.shape[0]
        _same_length("descriptors", self._descriptors, n)
        _same_length("landmarks", self._landmarks, n)
        assert self._state.shape[0] == n, "Unequal number of state and keypoints."
        return n

    def set_pose_for_new_tracks(self, pose: np.ndarray) -> None:
        """Start pose for every keypoint that begins a new track (features.py:224-237)."""
        new = self.state == 0
        assert np.all(np.isnan(self.poses[new]))
        assert pose.shape == (4, 4) or (
            pose.ndim == 3 and pose.shape[1:] == (4, 4) and pose.shape[0] == np.sum(new)
        ), "Invlaid shape for pose"
        self._poses[new] = pose

    def mask(self, mask: np.ndarray) -> None:
        """Keep the keypoints where `mask` is True (features.py:239-265)."""
        assert mask.shape == (self.length,), "Invalid mask shape"
        for name in _PER_POINT:
            value = getattr(self, name)
            if value is not None:
                setattr(self, name, value[mask])
