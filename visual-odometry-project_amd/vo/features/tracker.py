"""Tracker selection (reference: src/vo/features/tracker.py): the one entry point the
driver calls per frame."""
from vo.features.harris import HarrisCornerDetector
from vo.features.klt import KLTTracker
from vo.features.sift import SIFTDetector
from vo.primitives import Frame, Matches

_TRACKERS = {
    "klt": (KLTTracker, "track_features"),
    "harris": (HarrisCornerDetector, "featureMatcher"),
    "sift": (SIFTDetector, "get_sift_matches"),
}


class Tracker:
    def __init__(self, frame, mode="klt"):
        self._init_frame = frame
        self._mode = mode
        self._tracker = None
        self.initTracker(frame)

    def _entry(self):
        if self._mode not in _TRACKERS:
            raise Exception("Tracker Name not valid")                 # tracker.py:52,63
        return _TRACKERS[self._mode]

    def initTracker(self, frame: Frame) -> None:
        cls, _ = self._entry()
        self._tracker = cls(frame)

    def trackFeatures(self, curr_frame: Frame, new_frame: Frame) -> Matches:
        _, method = self._entry()
        return getattr(self._tracker, method)(curr_frame, new_frame)
