"""Oracle pinned against goldens the reference's own code produced at the sizes the path is
benchmarked at (tools/make_golden.py harris_full / harris_kitti / dlt_candidates):
  * full_harris.npz   -- HarrisCornerDetector(num_keypoints=2000) on one 1376x1241 frame
                         (the reference's own 2N-argmax loop, src/vo/features/harris.py:86-194);
  * kitti_harris.npz  -- frames 0 and 1 of the reference's KITTI fixture, 200 keypoints
                         (tests/test_harris.py:126-171);
  * dlt_candidates.npz -- triangulate_candidates with 2000 tracks, each with its own start pose
                         (src/vo/landmarks/triangulation.py:38-86)."""
import hashlib
import os
import sys

import numpy as np

from oracle import dlt_np, harris_np

G = os.path.join(os.path.dirname(__file__), "golden")


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def full_frame():
    from vo import synthetic
    g = np.load(os.path.join(G, "full_harris.npz"))
    img = synthetic.Stream(2, int(g["H"]), int(g["W"])).image(0)
    assert np.array_equal(_sha(img), g["image_sha256"]), "vo.synthetic no longer generates the frame the golden was made on"
    return g, img


def test_oracle_harris_at_configuration_size_matches_reference():
    g, img = full_frame()
    scores = harris_np.harris_scores(img, int(g["patch_size"]), float(g["kappa"]))
    assert np.array_equal(_sha(scores), g["scores_sha256"])
    assert np.array_equal(scores[600], g["scores_row600"])
    kp = harris_np.nms_keypoints_fast(scores, int(g["num_keypoints"]), int(g["nms_radius"]))
    assert np.array_equal(kp, g["keypoints"]), "the oracle's fast NMS walk differs from the reference's argmax loop"
    desc = harris_np.patch_descriptors(img, kp, int(g["descriptor_radius"]))
    assert np.array_equal(_sha(desc), g["descriptors_sha256"])


def test_oracle_harris_on_the_reference_kitti_fixture():
    g = np.load(os.path.join(G, "kitti_harris.npz"))
    for k in (0, 1):
        img = g["image%d" % k]
        assert img.shape == (370, 1226)
        scores = harris_np.harris_scores(img, 9, 0.09)
        assert np.array_equal(_sha(scores), g["scores%d_sha256" % k])
        kp = harris_np.nms_keypoints(scores, 200, 5)
        assert np.array_equal(kp, g["keypoints%d" % k])
        assert np.array_equal(harris_np.nms_keypoints_fast(scores, 200, 5), kp)
        desc = harris_np.patch_descriptors(img, kp, 9)
        assert np.array_equal(_sha(desc), g["descriptors%d_sha256" % k])
        assert np.array_equal(desc[:8], g["descriptors%d_head" % k])


def test_oracle_candidate_triangulation_with_per_track_poses():
    g = np.load(os.path.join(G, "dlt_candidates.npz"))
    m = g["mask"]
    poses = g["start_poses"][g["start_index"]]
    P1, P2 = dlt_np.candidate_projections(g["K"], poses[m], g["current_pose"])
    X = dlt_np.linear_triangulation(g["tracks"][m][:, :, 0], g["keypoints"][m][:, :, 0], P1, P2)
    assert X.shape == (2000, 3)
    assert np.allclose(X, g["X_cand"][:, :, 0], rtol=1e-9, atol=1e-9)
