"""Oracle (oracle/harris_np.py) pinned bit-exactly against goldens captured from
the reference's own code (tools/make_golden.py)."""
import glob
import hashlib
import os

import numpy as np
import pytest

from oracle import harris_np

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "harris_*.npz")))


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[7:-4] for p in CASES])
def test_oracle_harris_matches_reference_golden(path):
    g = np.load(path)
    img = g["image"]
    scores = harris_np.harris_scores(img, int(g["patch_size"]), float(g["kappa"]))
    assert scores.shape == img.shape
    assert np.array_equal(_sha(scores), g["scores_sha256"]), "response map differs from reference"
    if g["scores"].size:
        assert np.array_equal(scores, g["scores"])
    n, r = int(g["num_keypoints"]), int(g["nms_radius"])
    kp = harris_np.nms_keypoints(scores, n, r)
    assert np.array_equal(kp, g["keypoints"])
    kp_fast = harris_np.nms_keypoints_fast(scores, n, r)
    assert np.array_equal(kp_fast, g["keypoints"])
    desc = harris_np.patch_descriptors(img, kp, int(g["descriptor_radius"]))
    # the reference sizes the descriptor array by num_keypoints (harris.py:177,180)
    assert np.array_equal(_sha(desc), g["descriptors_sha256"])
    assert np.array_equal(desc[: len(g["descriptors"])], g["descriptors"])
