"""One image of the stream with its features (reference: src/vo/primitives/frame.py).
The reference's OpenCV debug windows (frame.py:38-76) are not part of the front-end."""
import numpy as np

from vo.primitives.features import Features
from vo.sensors import Camera


class Frame:
    def __init__(self, image: np.ndarray, features: Features = None, sensor: Camera = None,
                 intrinsics: np.ndarray = None):
        self.image = image
        self.frame_id = None
        self.features = features
        self.intrinsics = intrinsics
        self.sensor = sensor

    def get_frame_id(self) -> int:
        return self.frame_id

    def get_intrinsics(self) -> np.ndarray:
        return self.intrinsics

    def __repr__(self) -> str:
        return "Frame id: {}".format(self.frame_id)
