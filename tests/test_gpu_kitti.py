"""-m gpu: the headless driver on the reference's own KITTI test frames (tests/test_data/kitti/05/image_0/00000{0..5}.png,
held as arrays under tests/golden/) against the reference's ground truth (tests/test_data/kitti/poses/05.txt):
bootstrap on frames 0 and 2 (src/main.py:204-230), steady state on frames 3, 4, 5 (main.py:248-286).  SURVEY.md 8f-4.
Monocular: the translation's length is free, its direction and the rotation are not."""
import numpy as np
import pytest

from kitti_fixture import make_kitti_dir

pytestmark = pytest.mark.gpu


def errors_vs_ground_truth(traj, seq, frames):
    """(rotation error in degrees, cosine between the translation directions) of every trajectory entry, both relative
    to frame 0."""
    out = []
    T0 = np.linalg.inv(seq.ground_truth_pose(0))
    for T, k in zip(traj, frames):
        gt = T0 @ seq.ground_truth_pose(k)
        dR = T[:3, :3].T @ gt[:3, :3]
        ang = np.degrees(np.arccos(np.clip((np.trace(dR) - 1.0) / 2.0, -1.0, 1.0)))
        a, b = T[:3, 3], gt[:3, 3]
        cos = float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30)) if k > 0 else 1.0
        out.append((float(ang), cos))
    return out


def check(result, seq, what):
    frames = [0, 2, 3, 4, 5]
    traj = result["trajectory"]
    assert len(traj) == len(frames), what
    errs = errors_vs_ground_truth(traj, seq, frames)
    print(what, ["%.3f deg / cos %.5f" % e for e in errs], "landmarks", list(result["n_landmarks"]))
    for (ang, cos), k in zip(errs, frames):
        assert ang < 1.0, "%s: frame %d rotation off by %.3f degrees" % (what, k, ang)
        if k >= 2:
            assert cos > 0.99, "%s: frame %d translation direction cosine %.4f" % (what, k, cos)
    # the drive is forward and steady (0.56 m per frame): with the bootstrap's unit |t(0->2)| = 1 the camera moves on
    z = traj[:, 2, 3]
    assert np.all(np.diff(z) > 0.0), what


def test_device_pipeline_on_the_reference_kitti_frames(tmp_path):
    from vo import driver
    from vo.primitives import Sequence
    make_kitti_dir(str(tmp_path), frames=range(6))
    seq = Sequence("kitti", path=str(tmp_path))
    out = driver.run_on_device(seq, n_keypoints=1000, klt_win=17, klt_max_level=2, hyp=4000)
    check(out, seq, "run_on_device")
    assert len(out["results"]) == 3 and all(r.n_inliers >= 20 for r in out["results"])


@pytest.mark.parametrize("mode", ["klt", "harris", "sift"])
def test_class_driver_on_the_reference_kitti_frames(tmp_path, mode):
    """The reference's call order through the drop-in classes (Tracker mode klt / harris / sift, tracker.py:54-63)."""
    from vo import driver
    from vo.primitives import Sequence
    make_kitti_dir(str(tmp_path), frames=range(6))
    seq = Sequence("kitti", path=str(tmp_path))
    out = driver.run(seq, mode)
    check(out, seq, "run[%s]" % mode)
