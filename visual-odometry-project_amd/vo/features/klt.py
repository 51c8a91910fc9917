"""Shi-Tomasi corners + pyramidal Lucas-Kanade tracking (reference: src/vo/features/klt.py).
Corner map / candidates and the tracker are HIP kernels (vo_good_features,
vo_klt_track); the bookkeeping around them follows the reference."""
import sys

import numpy as np

from vo import _native
from vo.primitives import Features, Frame, Matches

TERM_CRITERIA_COUNT, TERM_CRITERIA_EPS = 1, 2


def _gray(image: np.ndarray) -> np.ndarray:
    if image.ndim == 2:
        return image
    b, g, r = (image[..., c].astype(np.int64) for c in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


class KLTTracker:
    # Shi-Tomasi parameters (klt.py:24-26)
    _feature_params = dict(maxCorners=500, qualityLevel=0.01, minDistance=8, blockSize=7)
    # Lucas-Kanade parameters (klt.py:29-33)
    _lk_params = dict(winSize=(17, 17), maxLevel=2, criteria=(TERM_CRITERIA_EPS | TERM_CRITERIA_COUNT, 10, 0.03))
    # points are kept only below this tracking error (klt.py:39)
    _error_threshold = 100
    # Extension for BASELINE.json configs[1] ("Harris+NMS -> KLT"): which detector find_corners runs.
    # "shi-tomasi" is the reference's cv2.goodFeaturesToTrack (klt.py:98); "harris" is the reference's own
    # HarrisCornerDetector.extractKeypoints (harris.py:86-158) with _harris_params.
    _detector = "shi-tomasi"
    _harris_params = dict(patch_size=9, kappa=0.09, num_keypoints=2000, nonmaximum_supression_radius=5)

    def __init__(self, frame, context=None):
        self._ctx = context
        self._min_inliers = 90
        self._num_features = None
        self._old_frame = frame
        self._frame = frame
        self._frame.features = Features(keypoints=self.find_corners(frame=self._frame))
        self._frame.features.uids = self._get_udis(self._frame.features.length)
        self._last_masks = []

    def _context(self):
        if self._ctx is None:
            self._ctx = _native.default_context()
        return self._ctx

    @property
    def frame(self) -> Frame:
        return self._frame

    @frame.setter
    def frame(self, frame: Frame) -> None:
        self._frame = frame
        self._frame.features = Features(self.find_corners(self._frame))

    @property
    def old_img_gray(self) -> np.ndarray:
        return _gray(self._old_frame.image)

    @property
    def img_gray(self) -> np.ndarray:
        return _gray(self._frame.image)

    def to_gray(self, img) -> np.ndarray:
        return _gray(img)

    def _get_udis(self, length: int) -> np.ndarray:
        return np.random.randint(0, np.iinfo(np.int32).max, size=length, dtype=np.int32)

    def _fill_udis(self, array: np.ndarray, target_length: int) -> np.ndarray:
        if array is None:
            return self._get_udis(target_length)
        return np.concatenate((array, self._get_udis(target_length - array.shape[0])))

    def find_corners(self, frame: Frame, mask=None, use_goodFeaturesToTrack=True) -> np.ndarray:
        """(n, 2, 1) float32 corners (klt.py:87-115)."""
        if not use_goodFeaturesToTrack:
            raise NotImplementedError("the cornerHarris + cornerSubPix branch (klt.py:99-110) is not on the "
                                      "front-end path and is not provided")
        if self._detector == "harris":
            h = self._harris_params
            pts = self._context().harris_keypoints(_gray(frame.image), h["patch_size"], h["kappa"], h["num_keypoints"],
                                                   h["nonmaximum_supression_radius"]).astype(np.float32)
        else:
            p = self._feature_params
            pts = self._context().good_features(_gray(frame.image), mask, p["maxCorners"], p["qualityLevel"],
                                                p["minDistance"], p["blockSize"])
        pts = pts.reshape((-1, 2, 1))
        self._num_features = pts.shape[0]
        return pts

    def update_features(self, new_keypoints: np.ndarray) -> Features:
        """Old frame's features extended by freshly detected keypoints (klt.py:117-189)."""
        old = self._old_frame.features
        k = new_keypoints.shape[0]
        feats = Features(keypoints=np.concatenate((old.keypoints, new_keypoints)),
                         landmarks=np.concatenate((old.landmarks, np.full((k, 3, 1), np.nan))))
        feats.state = np.concatenate((old.state, np.zeros(k)))
        feats.uids = self._fill_udis(old.uids, k + old.length)
        feats.tracks = np.concatenate((old.tracks, new_keypoints))
        feats.poses = np.concatenate((old.poses, np.stack([np.eye(4)] * k))) if k > 0 else old.poses
        feats.candidate_mask = np.concatenate((old.candidate_mask, np.zeros(k, dtype=bool)))
        return feats

    def track_features(self, curr_frame: Frame, new_frame: Frame) -> Matches:
        """Track the current frame's keypoints into the new frame; re-detect first when
        fewer than 80 % of the initial count survive (klt.py:191-280)."""
        self._old_frame = curr_frame
        self._frame = new_frame
        if self._old_frame.features is None or self._old_frame.features.length < self._num_features * 0.8:
            if sys.gettrace() is not None:
                print("Adding new features")
            # klt.py:216-222: the mask stays all-255 whenever the old frame has features (the
            # reference only draws exclusion circles in the branch where there are none)
            mask = np.ones_like(self.img_gray) * 255
            new_keypoints = self.find_corners(frame=self._old_frame, mask=mask)
            if self._old_frame.features is None:
                self._old_frame.features = Features(new_keypoints)
            else:
                self._old_frame.features = self.update_features(new_keypoints=new_keypoints)
            self._old_frame.features.uids = self._fill_udis(self._old_frame.features.uids,
                                                            self._old_frame.features.length)
        lk = self._lk_params
        crit = lk["criteria"]
        max_iter = crit[1] if crit[0] & TERM_CRITERIA_COUNT else 30
        eps = crit[2] if crit[0] & TERM_CRITERIA_EPS else 0.01
        next_pts, status, error = self._context().klt_track(
            self.old_img_gray, self.img_gray, self._old_frame.features.keypoints, win=lk["winSize"][0],
            max_level=lk["maxLevel"], max_iter=max_iter, eps=eps)
        next_pts = next_pts.reshape((-1, 2, 1))
        keep = np.logical_and(status.flatten().astype(bool), (error < self._error_threshold).flatten())
        if sys.gettrace() is not None:
            print(f"{np.sum(keep)/keep.shape[0]*100:.2f}% inliers")
        self.frame.features = Features(keypoints=next_pts)
        self.frame.features.uids = self._fill_udis(self._old_frame.features.uids, next_pts.shape[0])
        self.frame.features.mask(keep)
        self._old_frame.features.mask(keep)
        same = np.arange(0, self.frame.features.keypoints.shape[0]).reshape(-1, 1)
        self._matches = Matches(self._old_frame, self.frame, np.hstack((same, same)))
        return self._matches
