"""Sequence("kitti" / "parking") (reference: src/vo/primitives/loader.py:72-95, 160-229) on a directory laid out
like the reference's datasets, built here from the two frames, the calibration row and the ground-truth poses of
the reference's own KITTI test data that tests/golden/kitti_harris.npz holds as arrays."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


from kitti_fixture import make_kitti_dir  # noqa: E402


def test_kitti_sequence_reads_the_reference_layout(tmp_path):
    from vo.primitives import Sequence
    g = make_kitti_dir(str(tmp_path))
    seq = Sequence("kitti", path=str(tmp_path))
    assert len(seq) == 2
    K = seq.get_intrinsics()
    assert K.dtype == np.float32 and K.shape == (3, 3)
    assert np.array_equal(K, g["calib_P0"][:, :3].astype(np.float32))
    assert seq.get_camera().intrinsic_matrix is K
    frames = list(seq)
    assert [f.frame_id for f in frames] == [0, 1]
    for k, f in enumerate(frames):
        assert f.image.dtype == np.uint8 and f.image.shape == (370, 1226, 3)       # cv2.imread's 3 channels
        assert np.array_equal(f.image[:, :, 0], g["image%d" % k]) and np.array_equal(f.image[:, :, 2], g["image%d" % k])
        assert np.array_equal(f.intrinsics, K) and f.sensor.intrinsic_matrix is K or np.array_equal(f.sensor.intrinsic_matrix, K)
    T = seq.ground_truth_pose(1)
    assert T.shape == (4, 4) and np.allclose(T[:3], g["poses05_head"][1], atol=1e-6) and np.array_equal(T[3], [0, 0, 0, 1])
    with pytest.raises(StopIteration):
        next(seq)
    assert len(list(Sequence("kitti", path=str(tmp_path), increment=2))) == 1


def test_kitti_sequence_all_six_frames(tmp_path):
    from vo.primitives import Sequence
    g = make_kitti_dir(str(tmp_path), frames=range(6))
    seq = Sequence("kitti", path=str(tmp_path))
    assert len(seq) == 6
    frames = list(seq)
    assert [f.frame_id for f in frames] == list(range(6))
    for k, f in enumerate(frames):
        assert np.array_equal(f.image[:, :, 1], g["image%d" % k])
    # the car drives forward ~0.56 m per frame (poses/05.txt)
    z = [seq.ground_truth_pose(k)[2, 3] for k in range(6)]
    assert np.all(np.diff(z) > 0.5) and np.all(np.diff(z) < 0.6)


def test_parking_and_invalid_datasets(tmp_path):
    from PIL import Image
    from vo.primitives import Sequence
    d = tmp_path / "parking" / "images"
    d.mkdir(parents=True)
    img = (np.arange(48 * 64).reshape(48, 64) % 251).astype(np.uint8)
    Image.fromarray(img).save(str(d / "img_00000.png"))
    (tmp_path / "parking" / "K.txt").write_text("331.37, 0, 320,\n0, 369.568, 240,\n0, 0, 1,\n")
    seq = Sequence("parking", path=str(tmp_path))
    assert len(seq) == 1 and np.allclose(seq.get_intrinsics(), [[331.37, 0, 320], [0, 369.568, 240], [0, 0, 1]])
    assert np.array_equal(next(seq).image[:, :, 1], img)
    with pytest.raises(Exception, match="Invalid dataset"):
        Sequence("nowhere", path=str(tmp_path))
