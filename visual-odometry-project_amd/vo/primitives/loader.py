"""Frame source.  The reference's Sequence (src/vo/primitives/loader.py) reads the KITTI /
Malaga / parking datasets with cv2.imread; dataset I/O is outside the front-end, and this
drop-in iterates the deterministic synthetic stream of `vo.synthetic` instead, with the
same iterator / get_frame / get_camera surface (loader.py:184-242)."""
import numpy as np

from vo import synthetic
from vo.primitives.frame import Frame
from vo.sensors import Camera


class Sequence:
    def __init__(self, dataset: str = "synthetic", path: str = None, camera: int = 0, increment: int = 1,
                 n_frames: int = 50, height: int = 480, width: int = 640, seed: int = 2023, channels: int = 1):
        if dataset != "synthetic":
            raise NotImplementedError("only the in-repo synthetic stream is available (no dataset files are shipped)")
        self.dataset = dataset
        self.increment = increment
        self.n_frames, self.H, self.W, self.seed, self.channels = n_frames, height, width, seed, channels
        self.K = synthetic.intrinsics(height, width)
        self.idx = 0

    def __len__(self) -> int:
        return self.n_frames // self.increment

    def get_camera(self) -> Camera:
        return Camera(intrinsic_matrix=self.K)

    def ground_truth_pose(self, idx: int) -> np.ndarray:
        """Camera-to-world 4x4 of frame idx (analytic)."""
        return synthetic.pose_world_cam(idx * self.increment)

    def get_frame(self, idx: int) -> Frame:
        img, _, _, K = synthetic.render(idx * self.increment, self.H, self.W, self.seed)
        if self.channels == 3:
            img = np.repeat(img[:, :, None], 3, axis=2)
        frame = Frame(img, sensor=Camera(intrinsic_matrix=K), intrinsics=K)
        frame.frame_id = idx
        return frame

    def __iter__(self):
        return self

    def __next__(self) -> Frame:
        if self.idx >= len(self):
            raise StopIteration
        frame = self.get_frame(self.idx)
        self.idx += 1
        return frame
