"""SIFT detect + describe + brute-force matching (reference: src/vo/features/sift.py)."""
import numpy as np

from vo import _native
from vo.primitives import Features, Frame, Matches


class SIFTDetector:
    # cv2.SIFT_create() keeps every keypoint (nfeatures = 0, sift.py:10); an integer keeps the strongest that many
    # (BASELINE.json configs[2] caps the detector at 2000)
    _max_keypoints = None

    def __init__(self, frame: Frame, context=None):
        self._ctx = context
        kp, desc = self.detect_and_compute(frame=frame)
        frame.features = Features(keypoints=kp)
        frame.features.descriptors = desc

    def _context(self):
        if self._ctx is None:
            self._ctx = _native.default_context()
        return self._ctx

    def detect_and_compute(self, frame: Frame):
        """((n, 2, 1) keypoint positions, (n, 128) float32 descriptors) (sift.py:16-21)."""
        img = frame.image
        if img.ndim == 3:
            b, g, r = (img[..., c].astype(np.int64) for c in range(3))
            img = ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)
        kp, desc = self._context().sift(img, cap=type(self)._max_keypoints)
        return kp[:, :2].astype(np.float64).reshape(-1, 2, 1), desc

    def get_sift_matches(self, curr_frame: Frame, new_frame: Frame) -> Matches:
        """2-NN + ratio 0.8 + first-come uniqueness (sift.py:23-56)."""
        kp2, desc2 = self.detect_and_compute(new_frame)
        new_frame.features = Features(kp2)
        new_frame.features.descriptors = desc2
        good = self._context().match_knn2_ratio(curr_frame.features.descriptors, new_frame.features.descriptors, 0.8)
        if len(good) == 0:
            good = np.empty((0, 2), dtype=int)   # the reference passes a 1-D empty array here, which breaks Matches
        return Matches(curr_frame, new_frame, matches=good)
