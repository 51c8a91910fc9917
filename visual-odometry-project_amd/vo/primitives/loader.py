"""Frame source (reference: src/vo/primitives/loader.py).

`Sequence` keeps the reference's surface (constructor arguments, iterator, get_frame / get_intrinsics /
get_camera, loader.py:23-242) for

  * "kitti"    -- <data>/kitti/05/image_<camera>/*.png, intrinsics from line 2*camera+1 of calib.txt (the line the
                  reference reads, loader.py:86-93: 12 floats, float32, left 3x3 block), ground-truth poses from
                  <data>/kitti/poses/05.txt (12 floats per row, the format of the reference's test data);
  * "parking"  -- <data>/parking/images/*.png, intrinsics from K.txt (comma separated, loader.py:172-181);
  * "synthetic" -- the deterministic in-repo stream of `vo.synthetic` (no files), used by the tests and bench.py.

Images are read with PIL (OpenCV is not a dependency here) and handed on the way cv2.imread does: three equal
BGR channels for a grey PNG, uint8.  "malaga" (JPEG pairs + an ini-style calibration file, loader.py:97-158) is not
provided.  `path` is a directory, absolute or relative to the working directory: the reference locates it by
splitting its own file path on the project's name (loader.py:44-53), which only works inside its checkout."""
import glob
import os

import numpy as np

from vo import synthetic
from vo.primitives.frame import Frame
from vo.sensors import Camera


def _imread(path):
    from PIL import Image
    img = np.array(Image.open(path))
    if img.ndim == 2:                                  # cv2.imread's default flag: always 3 channels
        img = np.repeat(img[:, :, None], 3, axis=2)
    elif img.shape[2] >= 3:
        img = img[:, :, 2::-1]                         # RGB(A) -> BGR
    return np.ascontiguousarray(img, dtype=np.uint8)


class Sequence:
    def __init__(self, dataset: str = "synthetic", path: str = "./data", camera: int = 0, increment: int = 1,
                 rectified: bool = False, use_lowres: bool = False, n_frames: int = 50, height: int = 480,
                 width: int = 640, seed: int = 2023, channels: int = 1):
        self.dataset = dataset
        self._rel_data_path = path
        self.data_dir = os.path.abspath(path) if path is not None else None
        self.camera = camera
        self.increment = increment
        self.intrinsics = None
        self.idx = 0
        self.images = None
        self._gt = None
        if dataset == "synthetic":
            self.n_frames, self.H, self.W, self.seed, self.channels = n_frames, height, width, seed, channels
            self.intrinsics = synthetic.intrinsics(height, width)
        elif dataset == "kitti":
            self.images = self._load_kitti()
        elif dataset == "parking":
            self.images = self._load_parking()
        elif dataset == "malaga":
            raise NotImplementedError("the Malaga loader (loader.py:97-158) is not provided")
        else:
            raise Exception("Invalid dataset")                          # loader.py:69-70
        self.K = self.intrinsics
        self.sensor = Camera(intrinsic_matrix=self.intrinsics)

    # ---- datasets ----
    def _load_kitti(self):
        data_path = os.path.join(self.data_dir, "kitti", "05", "image_%d" % self.camera)
        image_paths = sorted(glob.glob(data_path + "/*.png"))
        with open(os.path.join(self.data_dir, "kitti", "05", "calib.txt"), "r") as f:
            line = f.readlines()[2 * self.camera + 1]                  # (sic) loader.py:88
        values = line.split(" ")[1:]
        values[-1] = values[-1].split("\n")[0]
        self.intrinsics = np.array([np.float32(v) for v in values]).reshape(3, 4)[:, :3]
        poses = os.path.join(self.data_dir, "kitti", "poses", "05.txt")
        if os.path.exists(poses):
            self._gt = np.loadtxt(poses).reshape(-1, 3, 4)
        return image_paths

    def _load_parking(self):
        data_path = os.path.join(self.data_dir, "parking", "images")
        image_paths = sorted(glob.glob(data_path + "/*.png"))
        with open(os.path.join(self.data_dir, "parking", "K.txt"), "r") as f:
            content = f.read().replace(" ", "").replace("\n", "")
        self.intrinsics = np.asarray([v for v in content.split(",") if v]).astype(np.float32).reshape(3, 3)
        poses = os.path.join(self.data_dir, "parking", "poses.txt")
        if os.path.exists(poses):
            self._gt = np.loadtxt(poses).reshape(-1, 3, 4)
        return image_paths

    # ---- the reference's surface ----
    def __len__(self) -> int:
        if self.dataset == "synthetic":
            return self.n_frames // self.increment
        return len(self.images)

    def get_intrinsics(self) -> np.ndarray:
        return self.intrinsics

    def get_camera(self) -> Camera:
        return self.sensor

    def ground_truth_pose(self, idx: int) -> np.ndarray:
        """Camera-to-world 4x4 of frame idx: analytic for the synthetic stream, the poses file otherwise."""
        if self.dataset == "synthetic":
            return synthetic.pose_world_cam(idx * self.increment)
        assert self._gt is not None, "no ground-truth poses for this sequence"
        return np.vstack([self._gt[idx], [0.0, 0.0, 0.0, 1.0]])

    def get_frame(self, idx: int) -> Frame:
        if self.dataset == "synthetic":
            img, _, _, K = synthetic.render(idx * self.increment, self.H, self.W, self.seed)
            if self.channels == 3:
                img = np.repeat(img[:, :, None], 3, axis=2)
        else:
            img, K = _imread(self.images[idx]), self.intrinsics
        frame = Frame(img, sensor=Camera(intrinsic_matrix=K), intrinsics=K)
        frame.frame_id = idx
        return frame

    def __iter__(self):
        return self

    def __next__(self) -> Frame:
        if self.dataset == "synthetic":
            if self.idx >= len(self):
                raise StopIteration
            frame = self.get_frame(self.idx)
            self.idx += 1
            return frame
        if self.idx >= len(self.images):                                # loader.py:219-229
            raise StopIteration
        frame = self.get_frame(self.idx)
        self.idx += self.increment
        return frame

    def __repr__(self) -> str:
        return "Sequence(dataset={}, path={}, camera={}, increment={})".format(self.dataset, self._rel_data_path,
                                                                              self.camera, self.increment)
