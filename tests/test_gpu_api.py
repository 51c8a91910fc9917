"""-m gpu: the drop-in `vo` classes (same names / kwargs as the reference) running on the
HIP library; includes the reference's own unit tests restated against this package."""
import os

import numpy as np
import pytest

from oracle import native
from scenarios import synthetic_image
from test_oracle_geometry import shift_image

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    from vo import _native
    c = _native.default_context()
    yield c


def cameras():
    from vo.sensors import Camera
    K = np.array([[500, 0, 320], [0, 500, 240], [0, 0, 1]], dtype=float)
    th1, th2 = np.pi / 8, np.pi / 32
    R = np.array([[np.cos(th1), -np.sin(th1), 0], [np.sin(th1), np.cos(th1), 0], [0, 0, 1]]) @ np.array(
        [[np.cos(th2), 0, np.sin(th2)], [0, 1, 0], [-np.sin(th2), 0, np.cos(th2)]])
    return (Camera(intrinsic_matrix=K, R=np.eye(3), t=np.zeros((3, 1))),
            Camera(intrinsic_matrix=K, R=R, t=np.array([[1.0, 1.0, -1.0]]).T))


# ---------------- matcher ----------------
@pytest.mark.parametrize("nq,nt,D,ratio", [(300, 280, 361, 0.85), (500, 700, 128, 0.8), (5, 3, 128, 0.8), (40, 1, 16, 0.9)])
def test_matcher_bytes_matches_oracle(ctx, nq, nt, D, ratio):
    rng = np.random.default_rng(nq + D)
    t = rng.integers(0, 256, size=(nt, D)).astype(np.float32)
    q = t[rng.integers(0, nt, size=nq)] + rng.integers(-12, 13, size=(nq, D))
    q = np.clip(q, 0, 255).astype(np.float32)
    q[::7] = rng.integers(0, 256, size=q[::7].shape)                      # unrelated queries
    q[1] = q[0]                                                           # duplicate query: uniqueness filter
    ref, _, _ = native.match_knn2_ratio(q, t, ratio)
    got = ctx.match_knn2_ratio(q, t, ratio)
    assert np.array_equal(got, ref)
    assert len(set(got[:, 1])) == len(got)


def test_matcher_float_path_matches_oracle(ctx):
    rng = np.random.default_rng(3)
    t = rng.normal(size=(150, 32)).astype(np.float32)
    q = (t[rng.integers(0, 150, size=120)] + rng.normal(scale=0.05, size=(120, 32))).astype(np.float32)
    ref, _, _ = native.match_knn2_ratio(q, t, 0.8)
    assert np.array_equal(ctx.match_knn2_ratio(q, t, 0.8), ref)
    assert ctx.match_knn2_ratio(np.zeros((0, 8), np.float32), t[:, :8], 0.8).shape == (0, 2)


# ---------------- Shi-Tomasi ----------------
@pytest.mark.parametrize("shape,seed", [((120, 160), 1), ((240, 320), 2), ((97, 131), 3)])
def test_good_features_match_oracle(ctx, shape, seed):
    img = synthetic_image(shape[0], shape[1], seed, block=9)
    assert np.array_equal(ctx.min_eigen_map(img, 7), native.min_eigen_map(img, 7))
    ref = native.good_features(img, None, 500, 0.01, 8, 7)
    got = ctx.good_features(img, None, 500, 0.01, 8, 7)
    assert np.array_equal(got, ref)
    d = np.linalg.norm(got[:, None] - got[None], axis=-1) + 1e9 * np.eye(len(got))
    assert d.min() >= 8
    mask = np.zeros(shape, np.uint8)
    mask[:, : shape[1] // 2] = 255
    refm = native.good_features(img, mask, 50, 0.05, 5, 5)
    gotm = ctx.good_features(img, mask, 50, 0.05, 5, 5)
    assert np.array_equal(gotm, refm) and np.all(gotm[:, 0] < shape[1] // 2)
    # a distance whose grid cells hold more candidates than the rounds kernel's cell lists: the one-workgroup walk decides
    assert np.array_equal(ctx.good_features(img, None, 0, 0.01, 40, 7), native.good_features(img, None, 0, 0.01, 40, 7))


@pytest.mark.parametrize("n,quality,min_dist,block", [(2000, 0.01, 8, 7), (500, 0.01, 7.5, 7), (0, 0.02, 12, 5),
                                                      (3000, 0.01, 0, 7), (100000, 0.001, 3, 3)])
def test_good_features_at_configuration_size(ctx, n, quality, min_dist, block):
    """cv2.goodFeaturesToTrack's ordering and greedy minimum-distance rule run on the device (descending radix sort,
    parallel rounds of the rule over all candidates -- or one workgroup walking it): the same corners in the same order
    as the oracle's sequential walk, at
    1376x1241, for the bootstrap's 2000 corners, a fractional distance, no corner limit, no distance, and a list
    long enough to need many blocks."""
    img = synthetic_image(1241, 1376, 17, block=9)
    ref = native.good_features(img, None, n, quality, min_dist, block)
    got = ctx.good_features(img, None, n, quality, min_dist, block)
    assert len(ref) > 400
    assert np.array_equal(got, ref)


# ---------------- the reference's tests/test_p3p.py, restated ----------------
def test_estimate_pose_like_reference_test(ctx):
    from vo.pose_estimation import P3PPoseEstimator
    from vo.primitives import Features
    cam1, cam2 = cameras()
    rng = np.random.default_rng(2023)
    landmarks = rng.uniform(-1, 1, size=(1000, 3, 1))
    landmarks[:, 2] = landmarks[:, 2] * 5 + 10
    points2 = cam2.project_points_world_frame(landmarks)
    est = P3PPoseEstimator(intrinsic_matrix=cam2.intrinsic_matrix, use_opencv=False, inlier_threshold=1,
                           outlier_ratio=0.9, confidence=0.99, max_iterations=1000)
    (R, t), inliers = est.estimate_pose(Features(keypoints=points2, landmarks=landmarks))
    assert R.shape == (3, 3) and t.shape == (3, 1) and inliers.shape == (1000,) and inliers.dtype == bool
    assert np.allclose(R, cam2.R, atol=1e-3) and np.allclose(t, cam2.t, atol=1e-3)   # tests/test_p3p.py:93-98
    # the estimator object keeps its generator and bounds between calls, like the reference's RANSAC
    n_it = est.ransac.n_iterations
    (R2, t2), _ = est.estimate_pose(Features(keypoints=points2, landmarks=landmarks))
    assert np.allclose(R2, cam2.R, atol=1e-3) and est.ransac.n_iterations <= n_it


def test_estimate_pose_equals_sequential_oracle(ctx):
    """Batched GPU hypotheses + replay == the reference's one-hypothesis-per-iteration loop."""
    from oracle import ransac_np
    from vo.pose_estimation import P3PPoseEstimator
    from vo.primitives import Features
    cam1, cam2 = cameras()
    rng = np.random.default_rng(7)
    X = rng.uniform(-1, 1, size=(600, 3, 1))
    X[:, 2] = X[:, 2] * 5 + 10
    x = cam2.project_points_world_frame(X) + rng.normal(0, 0.4, size=(600, 2, 1))
    x[:120] += rng.uniform(-30, 30, size=(120, 2, 1))
    K = cam2.intrinsic_matrix
    est = P3PPoseEstimator(intrinsic_matrix=K, use_opencv=False, inlier_threshold=1.0, max_iterations=1000,
                           nonlinear_refinement=False, batch_size=64)
    ref = ransac_np.p3p_ransac(X[:, :, 0], x[:, :, 0], K, 1.0, 0.9, 0.99, 1000)
    for _ in range(3):                                                    # persistent state across frames
        (R, t), inl = est.estimate_pose(Features(keypoints=x, landmarks=X))
        (Rr, tr), inl_r = ref.find_best_model(np.arange(600))
        assert np.array_equal(inl, inl_r)
        assert np.allclose(R, Rr, atol=1e-9) and np.allclose(t[:, 0], tr, atol=1e-9)
        assert est.ransac.n_iterations == ref.n_iterations and est.ransac.outlier_ratio == ref.outlier_ratio
        assert np.array_equal(est.ransac.rng.integers(0, 2**60, 3), ref.rng.integers(0, 2**60, 3))


# ---------------- the reference's tests/test_triangulation.py, restated ----------------
def test_linear_triangulation_and_relative_pose_like_reference_test(ctx):
    from vo.landmarks import LandmarksTriangulator
    cam1, cam2 = cameras()
    tri = LandmarksTriangulator(camera1=cam1, camera2=cam2, use_ransac=False, use_opencv=False)
    rng = np.random.default_rng(2023)
    for _ in range(3):
        X = rng.uniform(-1, 1, size=(1000, 3, 1))
        X[:, 2] = X[:, 2] * 5 + 10
        p1, p2 = cam1.project_points_world_frame(X), cam2.project_points_world_frame(X)
        ok = (np.all((0 <= p1) & (p1 <= 400), axis=-2) & np.all((0 <= p1) & (p2 <= 400), axis=-2)).flatten()
        X, p1, p2 = X[ok], p1[ok], p2[ok]
        M2, _ = tri._find_relative_pose(p1, p2)
        assert np.allclose(M2[:3, :3], cam2.R)
        assert np.allclose(M2[:3, 3:] / np.linalg.norm(M2[:3, 3:]), cam2.t / np.linalg.norm(cam2.t))
        M2[:3, 3:] *= np.linalg.norm(cam2.t) / np.linalg.norm(M2[:3, 3:])
        c2_T_w = np.vstack([M2, [0, 0, 0, 1]]) @ cam1.c_T_w
        Xt = tri._linear_triangulation(p1, p2, C1=cam1.intrinsic_matrix @ cam1.c_T_w[:3],
                                       C2=cam2.intrinsic_matrix @ c2_T_w[:3])
        assert np.allclose(X, Xt, atol=1e-4)                              # tests/test_triangulation.py:229


def test_triangulate_candidates_like_reference_test(ctx):
    from vo.landmarks import LandmarksTriangulator
    from vo.primitives import Features, Frame, Matches
    cam1, cam2 = cameras()
    tri = LandmarksTriangulator(camera1=cam1, camera2=cam2, use_ransac=False, use_opencv=False)
    rng = np.random.default_rng(5)
    X = rng.uniform(-1, 1, size=(800, 3, 1))
    X[:, 2] = X[:, 2] * 5 + 10
    p1, p2 = cam1.project_points_world_frame(X), cam2.project_points_world_frame(X)
    m = Matches(Frame(None, features=Features(keypoints=p1)), Frame(None, features=Features(keypoints=p2)),
                matches=np.stack([np.arange(len(p1))] * 2, axis=-1))
    m.frame2.features.candidate_mask = np.ones(len(p1), dtype=bool)
    assert np.all(m.frame2.features.tracks == p1) and np.all(m.frame2.features.poses == np.eye(4))
    Xt = tri.triangulate_candidates(m.frame2.features, np.linalg.inv(cam2.c_T_w))
    assert np.allclose(X, Xt, atol=1e-4)                                  # tests/test_triangulation.py:282


def same_up_to_scale(A, B, tol):
    """F and E are defined up to scale and sign: compared as unit Frobenius-norm matrices with the sign of B."""
    A, B = A / np.linalg.norm(A), B / np.linalg.norm(B)
    if np.sum(A * B) < 0:
        A = -A
    return np.abs(A - B).max() < tol


def same_candidate_set(M4, ref, tol):
    """_decompose_essential_matrix's four [R | +-T]: the same set; the order depends on the SVD routine's signs."""
    used = set()
    for m in M4:
        hit = [k for k in range(4) if k not in used and np.abs(m - ref[k]).max() < tol]
        if not hit:
            return False
        used.add(hit[0])
    return len(used) == 4


def test_bootstrap_matches_reference_golden(ctx):
    """The device bootstrap (csrc/bootstrap.hip) against the reference's own outputs (tests/golden/bootstrap.npz,
    use_opencv=False): 8-point F and E up to scale / sign, the four decompositions as a set, M, the landmarks; through
    RANSAC (the reference's sampler and sequential rule on the host, hypotheses + counts + closing fit on the device)
    the golden's inlier mask exactly."""
    from vo.landmarks import LandmarksTriangulator
    cam1, cam2 = cameras()
    g = np.load(os.path.join(G, "bootstrap.npz"))
    tri = LandmarksTriangulator(camera1=cam1, camera2=cam2, use_ransac=False, use_opencv=False, context=ctx)
    assert same_up_to_scale(tri._find_fundamental_matrix(g["x1"], g["x2"]), g["F"], 1e-9)
    assert same_up_to_scale(tri._find_essential_matrix(g["x1"], g["x2"]), g["E"], 1e-9)
    assert same_candidate_set(tri._decompose_essential_matrix(g["E"]), g["M4"], 1e-9)
    M, X = tri._find_relative_pose(g["x1"], g["x2"])
    assert np.allclose(M, g["M"], atol=1e-9) and np.allclose(X, g["X_tri"], rtol=1e-7, atol=1e-7)
    tri_r = LandmarksTriangulator(camera1=cam1, camera2=cam2, use_ransac=True, use_opencv=False, outlier_ratio=0.5,
                                  ransac_threshold=1e-3, ransac_confidence=0.99, context=ctx)
    Mr, Xr, inl = tri_r._find_relative_pose(g["x1"], g["x2_outliers"])
    assert np.array_equal(inl, g["inliers_ransac"]) and np.allclose(Mr, g["M_ransac"], atol=1e-9)
    assert np.allclose(Xr, g["X_ransac"], rtol=1e-6, atol=1e-6)


def test_bootstrap_kernels_match_the_oracle(ctx):
    """Hypotheses, counts and masks of a batch of samples, the closing fit and the relative pose against
    oracle/bootstrap_np.py (itself pinned to the reference's golden) on a noisy two-view scene with gross outliers: both
    error kinds, population-normalised and per-sample-normalised fits, at the bootstrap's size (2000 correspondences)."""
    from oracle import bootstrap_np as bo
    cam1, cam2 = cameras()
    rng = np.random.default_rng(11)
    n = 2000
    X = rng.uniform(-1, 1, size=(n, 3, 1))
    X[:, 2] = X[:, 2] * 5 + 10
    x1 = cam1.project_points_world_frame(X) + rng.normal(0, 0.2, size=(n, 2, 1))
    x2 = cam2.project_points_world_frame(X) + rng.normal(0, 0.2, size=(n, 2, 1))
    bad = rng.permutation(n)[:500]
    x2[bad] += rng.uniform(-50, 50, size=(500, 2, 1))
    samples = np.stack([rng.choice(n, size=8, replace=False) for _ in range(300)]).astype(np.int32)
    p1n, _ = bo.normalize_points(x1)
    p2n, _ = bo.normalize_points(x2)
    for (a, b, norm, kind, thr) in ((p1n, p2n, False, 0, 1e-4), (x1, x2, True, 1, 1.0)):
        F, counts, masks = ctx.fundamental_hypotheses(a, b, samples, thr, norm, kind, want_masks=True)
        pop = np.stack([a, b], axis=1)
        differ = 0
        for h in range(len(samples)):
            Fo = bo.find_fundamental_matrix(a[samples[h]], b[samples[h]], is_normalized=not norm)
            assert same_up_to_scale(F[h], Fo, 1e-7), h
            # decisions with the DEVICE's F through the oracle's error function: identical but for errors within rounding
            # of the threshold
            err = (bo.algebraic_errors if kind == 0 else bo.epipolar_errors)(F[h], pop)
            inl = err < thr
            near = np.abs(err - thr) < 1e-9 * thr
            assert np.array_equal(inl[~near], masks[h][~near]) and counts[h] == masks[h].sum()
            differ += int(near.sum())
        assert differ < 5
    # closing fit over a mask, both normalisations
    mask = np.ones(n, dtype=bool)
    mask[bad] = False
    assert same_up_to_scale(ctx.fundamental_fit(x1, x2, mask, normalize=True), bo.find_fundamental_matrix(x1[mask], x2[mask]), 1e-9)
    assert same_up_to_scale(ctx.fundamental_fit(p1n, p2n, mask, normalize=False),
                            bo.find_fundamental_matrix(p1n[mask], p2n[mask], True), 1e-9)
    # relative pose with and without an inlier mask
    K1, K2 = cam1.intrinsic_matrix, cam2.intrinsic_matrix
    Fm = bo.find_fundamental_matrix(x1[mask], x2[mask])
    for inl in (None, mask):
        M, Xd, m_out, M4 = ctx.relative_pose(x1, x2, K1, K2, Fm, inl)
        Mo, Xo, mo, M4o = bo.find_relative_pose(x1, x2, K1, K2, Fm, inl)
        assert same_candidate_set(M4, M4o, 1e-9) and np.allclose(M, Mo, atol=1e-9)
        assert np.array_equal(m_out, mo)
        ok = np.isfinite(Xo[:, :, 0]).all(axis=1)
        assert np.allclose(Xd[ok], Xo[ok][:, :, 0], rtol=1e-6, atol=1e-6)


# ---------------- trackers ----------------
def test_harris_feature_matcher(ctx):
    """tests/test_harris.py restated: types/shapes, plus recovery of a known shift."""
    from vo.features import HarrisCornerDetector, Tracker
    from vo.primitives import Frame, Matches
    prev, nxt = shift_image(240, 320, 11, 3.0, -2.0)
    det = HarrisCornerDetector(num_keypoints=200)
    m = det.featureMatcher(Frame(prev.copy()), Frame(nxt.copy()))
    assert isinstance(m, Matches)
    f1, f2 = m.frame1.features, m.frame2.features
    assert f1.keypoints.shape == f2.keypoints.shape == (200, 2, 1) and f1.descriptors.shape == (200, 361, 1)
    n = int((f2.state >= 1).sum())
    assert n >= 50 and n == int((f1.state >= 1).sum())
    flow = (f2.keypoints[:n] - f1.keypoints[:n])[:, :, 0]
    assert np.abs(np.median(flow[:, 0]) - 3) <= 1 and np.abs(np.median(flow[:, 1]) + 2) <= 1
    t = Tracker(Frame(prev.copy()), mode="harris")
    assert isinstance(t.trackFeatures(Frame(prev.copy()), Frame(nxt.copy())), Matches)


def test_klt_tracker(ctx):
    from vo.features import KLTTracker, Tracker
    from vo.primitives import Frame, Matches
    prev, nxt = shift_image(240, 320, 12, 2.5, 1.25)
    prev3 = np.repeat(prev[:, :, None], 3, axis=2)                       # the reference's KLT path takes 3 channels
    nxt3 = np.repeat(nxt[:, :, None], 3, axis=2)
    f0 = Frame(prev3)
    trk = KLTTracker(f0)
    n0 = f0.features.length
    assert 50 < n0 <= 500 and f0.features.keypoints.dtype == np.float32
    m = trk.track_features(f0, Frame(nxt3))
    assert isinstance(m, Matches)
    k1, k2 = m.frame1.features.keypoints, m.frame2.features.keypoints
    assert k1.shape == k2.shape and k1.shape[0] > 0.8 * n0
    flow = (k2 - k1)[:, :, 0]
    assert np.abs(np.median(flow[:, 0]) - 2.5) < 0.1 and np.abs(np.median(flow[:, 1]) - 1.25) < 0.1
    assert np.all(m.frame2.features.state == 1)
    t = Tracker(Frame(prev3.copy()), mode="klt")
    assert isinstance(t.trackFeatures(t._init_frame, Frame(nxt3)), Matches)


def test_headless_driver_runs_the_reference_call_sequence(ctx):
    """main.py's bootstrap + steady-state loop on the synthetic stream, KLT mode (3-channel
    frames as the reference's KLT path expects)."""
    from vo import driver
    from vo.primitives import Sequence
    seq = Sequence("synthetic", n_frames=14, height=480, width=640, channels=3)
    out = driver.run(seq, "klt")
    assert out["trajectory"].shape[1:] == (4, 4) and len(out["trajectory"]) == 13
    assert out["n_landmarks"].min() > 30
    err = driver.trajectory_error(out, seq)
    # forward motion 0.8 m/frame; monocular scale fitted once
    assert err["rms"] < 0.05 * err["path_length"], err


def test_harris_feature_matcher_on_the_reference_kitti_frames(ctx, tmp_path):
    """tests/test_harris.py:126-171 restated: Sequence("kitti") -> frames 0 and 1 -> HarrisCornerDetector(
    num_keypoints=200).featureMatcher; the reference asserts types / shapes / at least one match; here the
    keypoints are also the reference's own (tests/golden/kitti_harris.npz)."""
    from test_loader import make_kitti_dir
    from vo.features import HarrisCornerDetector
    from vo.primitives import Matches, Sequence
    g = make_kitti_dir(str(tmp_path))
    seq = Sequence("kitti", path=str(tmp_path))
    f1, f2 = seq.get_frame(0), seq.get_frame(1)
    harris = HarrisCornerDetector(f1, num_keypoints=200)
    m = harris.featureMatcher(f1, f2)
    assert isinstance(m, Matches)
    assert m.frame1.features.keypoints.shape[0] <= 200 and m.frame1.features.descriptors.shape[0] <= 200
    assert m.frame1.features.descriptors.shape == m.frame2.features.descriptors.shape
    assert len(m.frame1.features.keypoints) > 0
    # every keypoint of both frames is one of the reference's (Matches re-orders them)
    for feats, k in ((m.frame1.features, 0), (m.frame2.features, 1)):
        ref = {tuple(p) for p in g["keypoints%d" % k][:, :, 0]}
        assert {tuple(p) for p in feats.keypoints[:, :, 0]} == ref
    assert int((m.frame2.features.state >= 1).sum()) > 50


def test_headless_driver_on_the_device_pipeline(ctx):
    """Same bootstrap, steady state as the device-resident pipeline (vo.driver.run_on_device): images in,
    pose records out, Features / State never leave HBM."""
    from vo import driver
    from vo.primitives import Sequence
    seq = Sequence("synthetic", n_frames=24, height=480, width=640, channels=3)
    out = driver.run_on_device(seq, n_keypoints=500, context=ctx)
    assert out["trajectory"].shape[1:] == (4, 4) and len(out["trajectory"]) == 23
    assert out["n_landmarks"].min() > 30
    assert all(r.fault == 0 or r.recovered for r in out["results"])
    err = driver.trajectory_error(out, seq)
    assert err["rms"] < 0.05 * err["path_length"], err
    f = out["features"]
    assert f.length == out["results"][-1].n_tracked and int((f.state == 2).sum()) == out["results"][-1].n_landmarks


# ---------------- SIFT ----------------
@pytest.mark.parametrize("kind,shape,seed", [("smooth", (240, 320), 21), ("blocks", (200, 260), 5), ("smooth", (97, 131), 8)])
def test_sift_matches_oracle(ctx, kind, shape, seed):
    if kind == "smooth":
        img, _ = shift_image(shape[0], shape[1], seed, 0.0, 0.0)
    else:
        img = synthetic_image(shape[0], shape[1], seed, block=14, noise=3.0)
    kr, dr = native.sift(img)
    kg, dg = ctx.sift(img)
    assert len(kr) > 20
    assert kg.shape == kr.shape and np.array_equal(kg, kr), "SIFT keypoints not bit-identical to the oracle"
    assert np.array_equal(dg, dr), "SIFT descriptors not bit-identical to the oracle"
    assert np.all(np.diff(kg[:, 0]) >= 0) and np.all(dg == np.round(dg)) and dg.max() <= 255
    capped_r, _ = native.sift(img, cap=25)
    capped_g, dcg = ctx.sift(img, cap=25)
    assert np.array_equal(capped_g, capped_r) and len(capped_g) == min(25, len(kr))


def test_sift_tracker_mode(ctx):
    from vo.features import SIFTDetector, Tracker
    from vo.primitives import Frame, Matches
    a, b = shift_image(240, 320, 21, 4.0, -3.0)
    fa, fb = Frame(a), Frame(b)
    det = SIFTDetector(fa)
    assert fa.features.keypoints.shape[1:] == (2, 1) and fa.features.descriptors.shape[1] == 128
    m = det.get_sift_matches(fa, fb)
    assert isinstance(m, Matches)
    n = int((m.frame2.features.state >= 1).sum())
    flow = (m.frame2.features.keypoints[:n] - m.frame1.features.keypoints[:n])[:, :, 0]
    assert n > 60 and np.mean(np.abs(flow - [4, -3]).max(axis=1) < 1.0) > 0.85
    t = Tracker(Frame(a), mode="sift")
    assert isinstance(t.trackFeatures(t._init_frame, Frame(b)), Matches)
