#!/usr/bin/env python3
"""Capture golden vectors from the reference's own NumPy/SciPy code.

BUILD-CONTAINER ONLY: reads /root/reference in place (nothing is copied) and
refuses to run without it.  The reference imports ``cv2`` and ``pytransform3d``
at module scope; neither is installed, so EMPTY module objects are registered
for them (only the two integer constants klt.py evaluates in its class body are
set).  No OpenCV arithmetic is emulated: every code path captured below is pure
NumPy/SciPy inside the reference.

Writes tests/golden/*.npz (data only: inputs + expected outputs).
"""
import hashlib
import os
import sys
import types

import numpy as np

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")


def _import_reference():
    if not os.path.isdir(REF):
        sys.exit("make_golden.py: /root/reference is absent; goldens are committed, nothing to do")
    for name in ("cv2", "pytransform3d", "pytransform3d.camera",
                 "pytransform3d.transformations", "pytransform3d.plot_utils"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["cv2"].TERM_CRITERIA_EPS = 2
    sys.modules["cv2"].TERM_CRITERIA_COUNT = 1
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def golden_harris():
    from vo.features.harris import HarrisCornerDetector
    from vo.primitives import Frame
    from scenarios import synthetic_image

    cases = []
    # (name, image, kwargs)
    cases.append(("tex160", synthetic_image(120, 160, 11), dict(num_keypoints=50)))
    cases.append(("tex320", synthetic_image(240, 320, 12, block=12), dict(num_keypoints=200)))
    cases.append(("tex640", synthetic_image(480, 640, 13, block=16), dict(num_keypoints=500)))
    # r > patch_radius + 1: negative-slice no-op, same pixel repeats
    cases.append(("negslice", synthetic_image(96, 128, 14), dict(num_keypoints=12, nonmaximum_supression_radius=9, patch_size=5)))
    # near-flat image: exhausted-score tail of (0, 0)
    flat = np.full((64, 96), 90, np.uint8)
    flat[20:36, 30:50] = 200
    cases.append(("flat", flat, dict(num_keypoints=40)))
    # exact ties (symmetric squares): tie-break by lowest flat index
    sq = np.zeros((60, 80), np.uint8)
    sq[20:40, 30:50] = 200
    cases.append(("square", sq, dict(num_keypoints=8)))
    # other parameters
    cases.append(("p7k05", synthetic_image(100, 140, 15, block=6), dict(num_keypoints=60, patch_size=7, kappa=0.05, nonmaximum_supression_radius=3, descriptor_radius=4)))
    # saturated checkerboard: large responses, many ties
    yy, xx = np.indices((72, 88))
    cases.append(("checker", ((yy // 8 + xx // 8) % 2 * 255).astype(np.uint8), dict(num_keypoints=30)))
    # diagonal stripes: edges only, response clamps to 0 everywhere -> all (0, 0)
    cases.append(("stripes", ((yy + xx) // 8 % 2 * 255).astype(np.uint8), dict(num_keypoints=10)))

    for name, img, kw in cases:
        det = HarrisCornerDetector(**kw)
        # re-run the response exactly as extractKeypoints does to export the map:
        # the reference does not return it, so capture it through a subclass hook
        # on np.argmax-free path: call extractKeypoints and recompute the map by
        # calling the same lines through the public method on a copy.
        fr = det.extractKeypoints(Frame(img.copy()))
        fr = det.extractDescriptors(fr)
        kp = fr.features.keypoints
        desc = fr.features.descriptors
        scores = _reference_scores(det, img)
        np.savez_compressed(
            os.path.join(OUT, "harris_%s.npz" % name),
            image=img,
            patch_size=det._patch_size, kappa=det._kappa,
            num_keypoints=det._num_keypoints,
            nms_radius=det._nonmaximum_supression_radius,
            descriptor_radius=det._descriptor_radius,
            keypoints=kp, descriptors=desc[: min(16, len(desc))],
            descriptors_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(desc).tobytes()).digest(), np.uint8),
            scores=scores if scores.size <= 160 * 120 else np.zeros(0),
            scores_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(scores).tobytes()).digest(), np.uint8),
        )
        print("harris", name, img.shape, kw, "kp[0..3]=", kp[:3, :, 0].tolist())


def _reference_scores(det, img):
    """The response map as extractKeypoints builds it (harris.py:103-137): the
    method does not return it, so intercept the first np.argmax call it makes
    (harris.py:149), whose argument is the padded map before any suppression."""
    from vo.primitives import Frame
    import vo.features.harris as hmod

    captured = {}
    real = np.argmax

    class _Stop(Exception):
        pass

    def spy(a, *args, **kw):
        captured["scores"] = np.array(a, copy=True)
        raise _Stop()

    hmod.np.argmax = spy
    try:
        try:
            det.extractKeypoints(Frame(img.copy()))
        except _Stop:
            pass
    finally:
        hmod.np.argmax = real
    return captured["scores"]


def _test_cameras():
    from vo.sensors.camera import Camera
    K = np.array([[500, 0, 320], [0, 500, 240], [0, 0, 1]], dtype=float)
    th1, th2 = np.pi / 8, np.pi / 32
    R = np.array([[np.cos(th1), -np.sin(th1), 0], [np.sin(th1), np.cos(th1), 0], [0, 0, 1]])
    R = R @ np.array([[np.cos(th2), 0, np.sin(th2)], [0, 1, 0], [-np.sin(th2), 0, np.cos(th2)]])
    c1 = Camera(intrinsic_matrix=K, R=np.eye(3), t=np.zeros((3, 1)))
    c2 = Camera(intrinsic_matrix=K, R=R, t=np.array([[1.0, 1.0, -1.0]]).T)
    return c1, c2


def golden_dlt():
    from vo.landmarks.triangulation import LandmarksTriangulator
    from vo.primitives import Features

    c1, c2 = _test_cameras()
    tri = LandmarksTriangulator(camera1=c1, camera2=c2, use_ransac=False, use_opencv=False)
    rng = np.random.default_rng(2023)
    n = 300
    X = rng.uniform(-1, 1, size=(n, 3, 1))
    X[:, 2] = X[:, 2] * 5 + 10
    x1 = c1.project_points_world_frame(X)
    x2 = c2.project_points_world_frame(X)
    C1 = c1.intrinsic_matrix @ c1.c_T_w[:3]
    C2 = c2.intrinsic_matrix @ c2.c_T_w[:3]
    X_clean = tri._linear_triangulation(x1, x2, C1, C2)
    x1n = x1 + rng.normal(0, 0.5, size=x1.shape)
    x2n = x2 + rng.normal(0, 0.5, size=x2.shape)
    X_noisy = tri._linear_triangulation(x1n, x2n, C1, C2)

    # per-point start poses through triangulate_candidates (triangulation.py:38-86)
    feats = Features(keypoints=x2n.copy())
    poses = np.stack([np.eye(4)] * n)
    for i in range(n):
        a = 0.01 * (i % 7)
        poses[i, :3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        poses[i, :3, 3] = [0.05 * (i % 5), 0.0, 0.02 * (i % 3)]
    tracks = np.zeros((n, 2, 1))
    K = c1.intrinsic_matrix
    for i in range(n):
        Tcw = np.linalg.inv(poses[i])
        u = K @ (Tcw[:3, :3] @ X[i] + Tcw[:3, 3:])
        tracks[i] = u[:2] / u[2:]
    feats.tracks = tracks
    feats.poses = poses
    mask = np.ones(n, dtype=bool)
    mask[::9] = False
    feats.candidate_mask = mask
    cur_pose = np.linalg.inv(c2.c_T_w)
    X_cand = tri.triangulate_candidates(feats, cur_pose)
    np.savez_compressed(
        os.path.join(OUT, "dlt_cameras.npz"),
        K=K, C1=C1, C2=C2, X_true=X, x1=x1, x2=x2, X_clean=X_clean,
        x1n=x1n, x2n=x2n, X_noisy=X_noisy,
        cand_keypoints=x2n, cand_tracks=tracks, cand_poses=poses, cand_mask=mask,
        cand_current_pose=cur_pose, X_cand=X_cand,
    )
    print("dlt: clean err", np.abs(X_clean - X).max(), "cand n", X_cand.shape)


def golden_ransac():
    from vo.algorithms.ransac import RANSAC

    out = {}
    # sampler known answers (ransac.py:52, 92-94)
    for pop in (5, 30, 1000, 2000, 20000):
        rng = np.random.default_rng(2023)
        out["choice4_pop%d" % pop] = np.stack(
            [rng.choice(np.arange(pop), replace=False, size=4) for _ in range(64)])
    for pop, s in ((30, 3), (50, 8), (12000, 8), (100000, 4)):
        rng = np.random.default_rng(2023)
        out["choice%d_pop%d" % (s, pop)] = np.stack(
            [rng.choice(np.arange(pop), replace=False, size=s) for _ in range(32)])
    # iteration bound table (ransac.py:58-67)
    rows = []
    for conf in (0.9, 0.99, 0.999, 0.9999):
        for orat in (0.01, 0.1, 1 / 3, 0.5, 0.9, 0.99):
            for s in (3, 4, 8):
                r = RANSAC(s, np.zeros((10, 2)), None, None, 1.0, orat, conf)
                rows.append((conf, orat, s, r.compute_n_iterations()))
    out["n_iter_table"] = np.array(rows, dtype=np.float64)

    # full trace on the reference test's parabola problem (tests/test_ransac.py:9-72)
    rng = np.random.default_rng(2023)
    num_inliers, num_outliers, noise_ratio = 20, 10, 0.1
    poly = rng.uniform(size=[3, 1])
    extremum = -poly[1] / (2 * poly[0])
    xstart = extremum - 0.5
    lowest = np.polyval(poly, extremum)
    highest = np.polyval(poly, xstart)
    yspan = highest - lowest
    max_noise = noise_ratio * yspan
    x = rng.uniform(size=[1, num_inliers]) + xstart
    y = np.polyval(poly, x)
    y = y + (rng.uniform(size=y.shape) - 0.5) * 2 * max_noise
    data = np.concatenate([
        np.concatenate([x, rng.uniform(size=[1, num_outliers]) + xstart], axis=1),
        np.concatenate([y, rng.uniform(size=[1, num_outliers]) * yspan + lowest], axis=1),
    ], axis=0).T

    trace = {"idx": [], "n_inl": [], "n_iter": []}

    def model_fn(samples):
        trace["idx"].append(None)
        return np.polyfit(samples[:, 0], samples[:, 1], 2)

    def error_fn(p, pts):
        return np.abs(np.polyval(p, pts[:, 0]) - pts[:, 1])

    r = RANSAC(3, data, model_fn, error_fn, float(max_noise[0]) + 1e-5, 1 / 3, 0.99)
    out["parabola_n_iter0"] = np.array(r.n_iterations)
    model, inl = r.find_best_model()
    out["parabola_data"] = data
    out["parabola_poly"] = poly
    out["parabola_max_noise"] = max_noise
    out["parabola_model"] = model
    out["parabola_inliers"] = inl
    out["parabola_n_iter_final"] = np.array(r.n_iterations)
    out["parabola_outlier_ratio_final"] = np.array(r.outlier_ratio)
    out["parabola_model_calls"] = np.array(len(trace["idx"]))
    out["parabola_rng_next"] = r.rng.integers(0, 2**62, size=4)
    # a second call on the same object: state (rng, n_iterations, outlier_ratio) persists
    model2, inl2 = r.find_best_model()
    out["parabola_model2"] = model2
    out["parabola_inliers2"] = inl2
    out["parabola_n_iter_final2"] = np.array(r.n_iterations)
    np.savez_compressed(os.path.join(OUT, "ransac.npz"), **out)
    print("ransac: choice4_pop1000[:2]", out["choice4_pop1000"][:2].tolist(), "calls", out["parabola_model_calls"])


def golden_bookkeeping():
    import vo.primitives as P
    from vo.sensors.camera import Camera
    from vo.landmarks.triangulation import LandmarksTriangulator
    from scenarios import bookkeeping_scenario

    ns = types.SimpleNamespace(Features=P.Features, Frame=P.Frame, Matches=P.Matches,
                               State=P.State, Camera=Camera,
                               LandmarksTriangulator=LandmarksTriangulator)
    out = bookkeeping_scenario(ns)
    np.savez_compressed(os.path.join(OUT, "bookkeeping.npz"), **out)
    print("bookkeeping:", len(out), "arrays;",
          {k: int(v.sum()) for k, v in out.items() if k.endswith("post_candidate_mask")},
          {k: np.bincount(v.astype(int), minlength=3).tolist() for k, v in out.items() if k.endswith("post_state")})


def golden_bootstrap():
    """8-point / essential / cheirality outputs (triangulation.py:110-350), use_opencv=False."""
    from vo.landmarks.triangulation import LandmarksTriangulator
    from vo.helpers import normalize_points

    c1, c2 = _test_cameras()
    rng = np.random.default_rng(2023)
    n = 200
    X = rng.uniform(-1, 1, size=(n, 3, 1))
    X[:, 2] = X[:, 2] * 5 + 10
    x1 = c1.project_points_world_frame(X)
    x2 = c2.project_points_world_frame(X)
    tri = LandmarksTriangulator(camera1=c1, camera2=c2, use_ransac=False, use_opencv=False)
    F = tri._find_fundamental_matrix(x1, x2)
    E = tri._find_essential_matrix(x1, x2)
    M4 = tri._decompose_essential_matrix(E)
    M, Xt = tri._find_relative_pose(x1, x2)
    pn, T = normalize_points(x1)
    out = dict(x1=x1, x2=x2, X_true=X, F=F, E=E, M4=M4, M=M, X_tri=Xt, x1_norm=pn, T1=T,
               K=c1.intrinsic_matrix, R_true=c2.R, t_true=c2.t)
    # RANSAC route with 20% gross outliers
    x2o = x2.copy()
    bad = rng.permutation(n)[:40]
    x2o[bad] += rng.uniform(-60, 60, size=(40, 2, 1))
    tri_r = LandmarksTriangulator(camera1=c1, camera2=c2, use_ransac=True, use_opencv=False,
                                  outlier_ratio=0.5, ransac_threshold=1e-3, ransac_confidence=0.99)
    Mr, Xr, inl = tri_r._find_relative_pose(x1, x2o)
    out.update(x2_outliers=x2o, bad_idx=bad, M_ransac=Mr, X_ransac=Xr, inliers_ransac=inl)
    np.savez_compressed(os.path.join(OUT, "bootstrap.npz"), **out)
    print("bootstrap: |R-Rtrue|", np.abs(M[:, :3] - c2.R).max(), "ransac inliers", int(inl.sum()), "/", n,
          "|Rr-R|", np.abs(Mr[:, :3] - c2.R).max())


def golden_helpers():
    from vo.helpers import twist_to_H_matrix, H_matrix_to_twist
    rng = np.random.default_rng(5)
    tw = rng.normal(0, 0.3, size=(16, 6))
    Hs = np.stack([twist_to_H_matrix(t) for t in tw])
    back = np.stack([H_matrix_to_twist(H) for H in Hs])
    np.savez_compressed(os.path.join(OUT, "helpers.npz"), twists=tw, H=Hs, twists_back=np.real(back))
    print("helpers: roundtrip", np.abs(np.real(back) - tw).max())


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def _amd_synthetic():
    """This repository's synthetic stream generator, loaded by path (the name `vo` belongs to the
    reference while this script runs)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "amd_synthetic", os.path.join(ROOT, "visual-odometry-project_amd", "vo", "synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def golden_harris_full():
    """The reference's own extractKeypoints / extractDescriptors on one frame of BASELINE.json
    configs[1] (1376x1241, 2000 keypoints) -- ~5 s of its 2N-argmax loop.  The frame is frame 0 of
    vo.synthetic.Stream (regenerated by the test; only its SHA-256 is stored)."""
    from vo.features.harris import HarrisCornerDetector
    from vo.primitives import Frame
    syn = _amd_synthetic()
    img = syn.Stream(2, 1241, 1376).image(0)
    det = HarrisCornerDetector(num_keypoints=2000)
    fr = det.extractDescriptors(det.extractKeypoints(Frame(img.copy())))
    scores = _reference_scores(det, img)
    np.savez_compressed(os.path.join(OUT, "full_harris.npz"), image_sha256=_sha(img), H=1241, W=1376,
                        patch_size=det._patch_size, kappa=det._kappa, num_keypoints=2000,
                        nms_radius=det._nonmaximum_supression_radius, descriptor_radius=det._descriptor_radius,
                        keypoints=fr.features.keypoints, descriptors_sha256=_sha(fr.features.descriptors),
                        scores_sha256=_sha(scores), scores_row600=scores[600].copy())
    print("harris_full: kp[0..3]=", fr.features.keypoints[:3, :, 0].tolist())


def golden_harris_kitti():
    """tests/test_harris.py:126-171 (frames 0 and 1 of the reference's KITTI fixture, 200 keypoints):
    the part of featureMatcher that is the reference's own NumPy code (keypoints + descriptors of
    both frames; the match itself is cv2.BFMatcher and stays unpinned).  The two PNGs are data files
    of the reference's tests, read with PIL and stored as arrays."""
    from PIL import Image
    from vo.features.harris import HarrisCornerDetector
    from vo.primitives import Frame
    d = "/root/reference/tests/test_data/kitti/05/image_0"
    out = {}
    for k in (0, 1):
        img = np.array(Image.open(os.path.join(d, "%06d.png" % k)))
        assert img.dtype == np.uint8 and img.ndim == 2
        det = HarrisCornerDetector(num_keypoints=200)
        fr = det.extractDescriptors(det.extractKeypoints(Frame(img.copy())))
        out["image%d" % k] = img
        out["keypoints%d" % k] = fr.features.keypoints
        out["descriptors%d_sha256" % k] = _sha(fr.features.descriptors)
        out["descriptors%d_head" % k] = fr.features.descriptors[:8]
        out["scores%d_sha256" % k] = _sha(_reference_scores(det, img))
    with open("/root/reference/tests/test_data/kitti/05/calib.txt") as f:
        out["calib_P0"] = np.array(f.readline().split()[1:], dtype=np.float64).reshape(3, 4)
    out["poses05_head"] = np.loadtxt("/root/reference/tests/test_data/kitti/poses/05.txt", max_rows=6).reshape(-1, 3, 4)
    np.savez_compressed(os.path.join(OUT, "kitti_harris.npz"), **out)
    print("harris_kitti:", out["image0"].shape, "kp0[0..3]=", out["keypoints0"][:3, :, 0].tolist())


def golden_kitti_frames():
    """Frames 2..5 of the reference's KITTI test data (tests/test_data/kitti/05/image_0; frames 0 and 1 are in
    kitti_harris.npz with the calibration row and the first six ground-truth poses): with them the headless driver runs
    bootstrap (frames 0, 2) + three steady-state frames on real images against poses/05.txt.  Data files
    of the reference's tests, read with PIL and stored as arrays; no reference code is involved."""
    from PIL import Image
    d = "/root/reference/tests/test_data/kitti/05/image_0"
    out = {}
    for k in (2, 3, 4, 5):
        img = np.array(Image.open(os.path.join(d, "%06d.png" % k)))
        assert img.dtype == np.uint8 and img.shape == (370, 1226)
        out["image%d" % k] = img
    np.savez_compressed(os.path.join(OUT, "kitti_frames.npz"), **out)
    print("kitti_frames:", sorted(out))


def golden_dlt_candidates():
    """triangulate_candidates (triangulation.py:38-86) with 2000 tracks, each with its own start pose
    (proj1[i] = K inv(pose_start_i)[:3]) -- the shape the per-frame loop calls it at (main.py:279-283)."""
    from vo.landmarks.triangulation import LandmarksTriangulator
    from vo.primitives import Features
    c1, c2 = _test_cameras()
    tri = LandmarksTriangulator(camera1=c1, camera2=c2, use_ransac=False, use_opencv=False)
    rng = np.random.default_rng(4046)
    n = 2200
    K = c1.intrinsic_matrix
    X = rng.uniform(-3, 3, size=(n, 3, 1))
    X[:, 2] = X[:, 2] * 4 + 18
    # 37 distinct start poses (a drive: forward motion + yaw), assigned to tracks at random
    starts = []
    for k in range(37):
        a = 0.01 * np.sin(0.1 * k)
        T = np.eye(4)
        T[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        T[:3, 3] = [0.02 * k, 0.0, 0.12 * k]
        starts.append(T)
    starts = np.stack(starts)
    which = rng.integers(0, 37, size=n)
    poses = starts[which]
    cur = np.eye(4)
    a = 0.015
    cur[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    cur[:3, 3] = [0.9, -0.02, 5.5]

    def proj(T, Xw):
        Tcw = np.linalg.inv(T)
        u = K @ (Tcw[:3, :3] @ Xw + Tcw[:3, 3:])
        return u[:2] / u[2:]

    tracks = np.stack([proj(poses[i], X[i]) for i in range(n)]) + rng.normal(0, 0.3, size=(n, 2, 1))
    kps = np.stack([proj(cur, X[i]) for i in range(n)]) + rng.normal(0, 0.3, size=(n, 2, 1))
    feats = Features(keypoints=kps.copy())
    feats.tracks = tracks
    feats.poses = poses
    mask = np.ones(n, dtype=bool)
    mask[rng.permutation(n)[:200]] = False
    feats.candidate_mask = mask
    Xc = tri.triangulate_candidates(feats, cur)
    np.savez_compressed(os.path.join(OUT, "dlt_candidates.npz"), K=K, keypoints=kps, tracks=tracks,
                        start_poses=starts, start_index=which, mask=mask, current_pose=cur, X_true=X, X_cand=Xc)
    print("dlt_candidates:", Xc.shape, "median err", float(np.median(np.abs(Xc - X[mask]))))


if __name__ == "__main__":
    _import_reference()
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["harris", "dlt", "ransac", "bookkeeping", "bootstrap", "helpers", "harris_full",
                             "harris_kitti", "dlt_candidates", "kitti_frames"]
    for w in which:
        globals()["golden_" + w]()
    # (python tools/make_golden.py kitti_frames: data files only)
