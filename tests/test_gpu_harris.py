"""-m gpu: HIP Harris response / greedy NMS / descriptors through the C ABI,
bit-exact against the reference goldens and the NumPy oracle."""
import glob
import hashlib
import os

import numpy as np
import pytest

from oracle import harris_np
from scenarios import synthetic_image

pytestmark = pytest.mark.gpu

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "harris_*.npz")))


@pytest.fixture(scope="module")
def ctx():
    from vo import _native
    c = _native.Context(0)
    yield c
    c.close()


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[7:-4] for p in CASES])
def test_harris_matches_reference_golden(ctx, path):
    g = np.load(path)
    img = g["image"]
    n, r = int(g["num_keypoints"]), int(g["nms_radius"])
    kp, scores = ctx.harris_keypoints(img, int(g["patch_size"]), float(g["kappa"]), n, r, want_scores=True)
    assert np.array_equal(_sha(scores), g["scores_sha256"]), "response map not bit-identical to the reference"
    assert np.array_equal(kp.reshape(n, 2, 1), g["keypoints"]), "keypoints differ from the reference"
    desc = ctx.patch_descriptors(img, kp, int(g["descriptor_radius"]))
    assert np.array_equal(_sha(desc.reshape(n, -1, 1)), g["descriptors_sha256"])


@pytest.mark.parametrize("shape,seed,block,n,r,patch", [
    ((97, 131), 1, 5, 64, 5, 9),
    ((240, 320), 2, 8, 300, 5, 9),
    ((480, 640), 3, 10, 500, 5, 9),
    ((333, 517), 4, 3, 1000, 3, 7),
    ((200, 200), 5, 16, 2000, 5, 9),      # more keypoints than corners: (0,0) tail
    ((130, 170), 6, 6, 150, 8, 5),        # r > patch/2+1: repeated-pixel rule
    ((64, 64), 7, 4, 16, 0, 3),           # r = 0
    ((17, 300), 8, 4, 20, 5, 9),          # thin images
    ((300, 19), 9, 4, 20, 5, 9),
    ((201, 333), 10, 6, 400, 5, 9),       # interior tiles of the packed kernel with rows at every byte alignment
    ((150, 270), 11, 1, 300, 5, 9),       # per-pixel noise: dense maxima
])
def test_harris_matches_oracle_seeded(ctx, shape, seed, block, n, r, patch):
    img = synthetic_image(shape[0], shape[1], seed, block=block)
    ref_scores = harris_np.harris_scores(img, patch, 0.09)
    ref_kp = harris_np.nms_keypoints_fast(ref_scores, n, r)
    kp, scores = ctx.harris_keypoints(img, patch, 0.09, n, r, want_scores=True)
    assert np.array_equal(scores, ref_scores)
    assert np.array_equal(kp, ref_kp[:, :, 0])


@pytest.mark.parametrize("kind", ["checker1", "checker3", "stripes", "edges"])
def test_response_extreme_gradients(ctx, kind):
    """0 / 255 patterns drive the Sobel sums to +-1020 and the box sums to their maxima: the packed 16-bit
    gradients and the dot-product sums of the response kernel must still be exact."""
    H, W = 141, 275
    y, x = np.mgrid[0:H, 0:W]
    if kind == "checker1":
        img = ((x + y) & 1) * 255
    elif kind == "checker3":
        img = (((x // 3) + (y // 3)) & 1) * 255
    elif kind == "stripes":
        img = ((x // 2) & 1) * 255
    else:
        img = np.where((x % 37 < 18) ^ (y % 29 < 14), 255, 0)
    img = img.astype(np.uint8)
    ref = harris_np.harris_scores(img, 9, 0.09)
    got = ctx.harris_response(img, 9, 0.09)
    assert np.array_equal(got, ref)


def test_nms_plateaus_and_ties(ctx):
    """Quantised score maps: large plateaus of equal scores exercise the index
    tie-break and long greedy dependency chains (the A1 path)."""
    rng = np.random.default_rng(42)
    for q, shape, n, r in [(4, (120, 160), 200, 5), (2, (90, 90), 400, 2), (16, (200, 300), 3000, 4)]:
        sc = rng.integers(0, q, size=shape).astype(np.float64) * 1.5
        sc[rng.random(shape) < 0.3] = 0.0
        ref = harris_np.nms_keypoints_fast(sc, n, r)[:, :, 0]
        got = ctx.nms_keypoints(sc, n, r)
        assert np.array_equal(got, ref), (q, shape, n, r)
    # constant map: pure chain, row-major picks every r+1
    sc = np.full((40, 50), 3.0)
    ref = harris_np.nms_keypoints(sc, 30, 3)[:, :, 0]
    got = ctx.nms_keypoints(sc, 30, 3)
    assert np.array_equal(got, ref)


def test_nms_smooth_ridges(ctx):
    """Smooth maps have few strict maxima and long monotone ridges."""
    yy, xx = np.mgrid[0:150, 0:200]
    sc = np.maximum(0, np.sin(xx / 9.0) * np.cos(yy / 7.0) + 0.3 * np.sin((xx + yy) / 23.0)) * 1e6
    for n, r in [(50, 5), (400, 5), (2000, 2)]:
        ref = harris_np.nms_keypoints_fast(sc, n, r)[:, :, 0]
        got = ctx.nms_keypoints(sc, n, r)
        assert np.array_equal(got, ref), (n, r)


def test_full_size_properties(ctx):
    """BASELINE config sizes: checked against the oracle's fast walk (the
    reference's own loop needs minutes here) and by size-independent properties."""
    H, W, N, r = 1241, 1376, 2000, 5
    img = synthetic_image(H, W, 99, block=12)
    kp, scores = ctx.harris_keypoints(img, 9, 0.09, N, r, want_scores=True)
    ref_scores = harris_np.harris_scores(img, 9, 0.09)
    assert np.array_equal(scores, ref_scores)
    x, y = kp[:, 0].astype(int), kp[:, 1].astype(int)
    s = scores[y, x]
    assert np.all(np.diff(s) <= 0), "picks must come in non-increasing score order"
    d = np.maximum(np.abs(x[:, None] - x[None]), np.abs(y[:, None] - y[None]))
    d[np.arange(N), np.arange(N)] = 99
    assert d.min() > r, "two picks inside one suppression window"
    ref_kp = harris_np.nms_keypoints_fast(ref_scores, N, r)[:, :, 0]
    assert np.array_equal(kp, ref_kp)


def test_configuration_size_matches_the_reference_itself(ctx):
    """One 1376x1241 frame, 2000 keypoints (BASELINE.json configs[1]) against what the reference's own
    extractKeypoints / extractDescriptors produced on it (tests/golden/full_harris.npz)."""
    from test_oracle_reference_size import full_frame
    g, img = full_frame()
    n, r = int(g["num_keypoints"]), int(g["nms_radius"])
    kp, scores = ctx.harris_keypoints(img, int(g["patch_size"]), float(g["kappa"]), n, r, want_scores=True)
    assert np.array_equal(_sha(scores), g["scores_sha256"]), "response map not bit-identical to the reference"
    assert np.array_equal(kp.reshape(n, 2, 1), g["keypoints"]), "keypoints differ from the reference"
    desc = ctx.patch_descriptors(img, kp, int(g["descriptor_radius"]))
    assert np.array_equal(_sha(desc.reshape(n, -1, 1)), g["descriptors_sha256"])


def test_reference_kitti_fixture(ctx):
    """Frames 0 and 1 of the reference's own KITTI test data (tests/test_harris.py:126-171), 200 keypoints."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "kitti_harris.npz"))
    for k in (0, 1):
        img = g["image%d" % k]
        kp, scores = ctx.harris_keypoints(img, 9, 0.09, 200, 5, want_scores=True)
        assert np.array_equal(_sha(scores), g["scores%d_sha256" % k])
        assert np.array_equal(kp.reshape(200, 2, 1), g["keypoints%d" % k])
        desc = ctx.patch_descriptors(img, kp, 9)
        assert np.array_equal(_sha(desc.reshape(200, -1, 1)), g["descriptors%d_sha256" % k])


@pytest.mark.parametrize("shape,S,n", [((240, 320), 3, 300), ((97, 131), 5, 64), ((480, 640), 2, 500)])
def test_batched_detector_equals_single_calls(ctx, shape, S, n):
    """Several sequences per launch (grid's extra dimension = sequence): S different frames through one set of
    launches give what S single calls give, score maps and keypoints bit for bit -- including after a call with
    a different S on the same context (workspace regrown, histograms / state maps re-zeroed)."""
    imgs = np.stack([synthetic_image(shape[0], shape[1], 40 + q, block=6 + q) for q in range(S)])
    imgs[-1, 10:40, 10:60] = 128                              # one frame with a flat patch: plateaus / ties
    kp, sc = ctx.harris_keypoints_batch(imgs, 9, 0.09, n, 5, want_scores=True)
    for q in range(S):
        kq, sq = ctx.harris_keypoints(imgs[q], 9, 0.09, n, 5, want_scores=True)
        assert np.array_equal(sc[q], sq) and np.array_equal(kp[q], kq), q
    kp2 = ctx.harris_keypoints_batch(imgs[:1], 9, 0.09, n, 5)
    assert np.array_equal(kp2[0], kp[0])
    kp3 = ctx.harris_keypoints_batch(imgs, 9, 0.09, n, 5)
    assert np.array_equal(kp3, kp)


def test_errors_are_reported_not_raised_across_abi(ctx):
    from vo._native import VoError
    img = np.zeros((32, 32), np.uint8)
    with pytest.raises(VoError):
        ctx.harris_response(img, patch=4)
    with pytest.raises(VoError):
        ctx.harris_keypoints(img, num_keypoints=0)
    with pytest.raises(VoError):
        ctx.nms_keypoints(np.zeros((8, 8)), 4, 99)
