"""Harris corners + raw patch descriptors + brute-force matching
(reference: src/vo/features/harris.py).  Response map, exact greedy NMS, patch gather
and the 2-NN distance search are HIP kernels (vo_harris_keypoints,
vo_patch_descriptors, vo_match_knn2_ratio)."""
import numpy as np

from vo import _native
from vo.primitives import Features, Frame, Matches


def _gray(image: np.ndarray) -> np.ndarray:
    """2-D images pass through; 3-channel input is reduced with the BT.601 weights
    cv2.cvtColor(BGR2GRAY) uses (harris.py:36-48)."""
    if image.ndim == 2:
        return image
    b, g, r = (image[..., c].astype(np.int64) for c in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


class HarrisCornerDetector:
    # Tracker(frame, mode="harris") constructs the detector with its defaults (tracker.py:43-50: 1000 keypoints); a
    # caller that wants another count there (BASELINE.json's configurations name 2000) sets this before
    _num_keypoints_override = None

    def __init__(self, frame: Frame = None, patch_size: int = 9, kappa: float = 0.09, num_keypoints: int = 1000,
                 nonmaximum_supression_radius: int = 5, descriptor_radius: int = 9, match_lambda: float = 4.0,
                 context=None):
        self._frame1 = frame
        self._frame2 = frame
        self._patch_size = patch_size
        self._kappa = kappa
        self._num_keypoints = type(self)._num_keypoints_override or num_keypoints
        self._nonmaximum_supression_radius = nonmaximum_supression_radius
        self._descriptor_radius = descriptor_radius
        self._match_lambda = match_lambda
        self._ctx = context

    def _context(self):
        if self._ctx is None:
            self._ctx = _native.default_context()
        return self._ctx

    @property
    def img1_gray(self) -> np.ndarray:
        return _gray(self._frame1.image)

    @property
    def img2_gray(self) -> np.ndarray:
        return _gray(self._frame2.image)

    def featureMatcher(self, curr_frame: Frame, new_frame: Frame) -> Matches:
        """Detect + describe on the new frame (and on the current one if it has no features
        yet), then match (harris.py:50-84)."""
        self._frame1, self._frame2 = curr_frame, new_frame
        self._frame1.image = self.img1_gray
        self._frame2.image = self.img2_gray
        if self._frame1.features is None:
            self._frame1 = self.extractDescriptors(self.extractKeypoints(self._frame1))
        self._frame2 = self.extractDescriptors(self.extractKeypoints(self._frame2))
        return self.matchDescriptor(self._frame1, self._frame2)

    def extractKeypoints(self, frame: Frame) -> Frame:
        """num_keypoints strongest Harris corners under greedy box NMS (harris.py:86-158)."""
        kp = self._context().harris_keypoints(_gray(frame.image), self._patch_size, self._kappa,
                                              self._num_keypoints, self._nonmaximum_supression_radius)
        assert frame.features is None, "Frame already has features"
        frame.features = Features(kp.reshape(-1, 2, 1))
        return frame

    def extractDescriptors(self, frame: Frame) -> Frame:
        """(N, (2r+1)^2, 1) raw patches around the keypoints (harris.py:160-194)."""
        kp = frame.features.keypoints
        desc = self._context().patch_descriptors(_gray(frame.image), kp[:, :, 0], self._descriptor_radius)
        frame.features.descriptors = desc.reshape(kp.shape[0], -1, 1)
        return frame

    def matchDescriptor(self, frame1: Frame, frame2: Frame) -> Matches:
        """2-NN + ratio 0.85 + first-come uniqueness (harris.py:196-264)."""
        d1 = frame1.features.descriptors.astype(np.float32)
        d2 = frame2.features.descriptors.astype(np.float32)
        pairs = self._context().match_knn2_ratio(d1.reshape(len(d1), -1), d2.reshape(len(d2), -1), 0.85)
        m = pairs if len(pairs) > 0 else np.empty(shape=(0, 2), dtype=int)
        return Matches(frame1, frame2, m)
