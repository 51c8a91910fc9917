"""The reference's KITTI test data (tests/test_data/kitti/05: six frames, calib.txt, poses/05.txt rows 0..5) as a
directory laid out the way the reference's loader reads it (src/vo/primitives/loader.py:72-95), rebuilt from the arrays
tests/golden/kitti_harris.npz (frames 0, 1, calibration, poses) and tests/golden/kitti_frames.npz (frames 2..5) hold."""
import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), "golden")


def make_kitti_dir(root, frames=(0, 1)):
    from PIL import Image
    g = dict(np.load(os.path.join(G, "kitti_harris.npz")))
    if any(k > 1 for k in frames):
        g.update(np.load(os.path.join(G, "kitti_frames.npz")))
    d = os.path.join(root, "kitti", "05", "image_0")
    os.makedirs(d)
    os.makedirs(os.path.join(root, "kitti", "poses"))
    for k in frames:
        Image.fromarray(g["image%d" % k]).save(os.path.join(d, "%06d.png" % k))
    P0 = g["calib_P0"]
    P1 = P0.copy()
    P1[0, 3] = -379.8145                                   # (the right camera's row differs in its 4th column only)
    with open(os.path.join(root, "kitti", "05", "calib.txt"), "w") as f:
        for name, P in (("P0", P0), ("P1", P1), ("P2", P0), ("P3", P0)):
            f.write(name + ": " + " ".join("%.12e" % v for v in P.reshape(-1)) + "\n")
    np.savetxt(os.path.join(root, "kitti", "poses", "05.txt"), g["poses05_head"].reshape(-1, 12), fmt="%.6e")
    return g
