"""Headless driver: the call sequence of the reference's ``main()`` (src/main.py:168-330)
without its matplotlib panes, ``time.sleep`` and dataset files.

    bootstrap on frames 0 and 2 (main.py:204-230):
        trackFeatures -> update_from_matches -> triangulate_matches -> outlier / inlier mask
        plumbing -> update_with_local_pose -> update_with_local_landmarks -> reset_outliers
    every later frame (main.py:248-286):
        trackFeatures -> estimate_pose(Features(triangulated keypoints, landmarks)) ->
        outliers[triangulate_inliers] = ~inliers -> update_from_matches ->
        update_with_world_pose -> reset_outliers -> compute_candidates ->
        triangulate_candidates -> update_with_world_landmarks

The monocular bootstrap leaves the scale free (|t| = 1 between frames 0 and 2), so the
trajectory is compared with ground truth after a single global scale fit.
"""
import time

import numpy as np

from vo.features import Tracker
from vo.landmarks import LandmarksTriangulator
from vo.pose_estimation import P3PPoseEstimator
from vo.primitives import Features, Sequence, State


def make_estimators(camera, ransac_threshold=0.25):
    """The triangulator and pose estimator as src/main.py:185-201 configures them (ransac_threshold: 0.25 px there)."""
    triangulator = LandmarksTriangulator(camera1=camera, camera2=camera, use_ransac=True, use_opencv=True,
                                         outlier_ratio=0.9, ransac_threshold=ransac_threshold, ransac_confidence=0.999)
    pose_estimator = P3PPoseEstimator(use_opencv=True, intrinsic_matrix=camera.intrinsic_matrix,
                                      inlier_threshold=1.25, outlier_ratio=0.9, confidence=0.9999,
                                      nonlinear_refinement=True)
    return triangulator, pose_estimator


def bootstrap(sequence: Sequence, tracker_mode: str = "klt", tracker_setup=None, ransac_threshold=0.25):
    """main.py:204-230: frames 0 and 2 -> (state, tracker, triangulator, pose_estimator).  The 8-point RANSAC's
    hypotheses, counts and closing fit, the essential-matrix decomposition and the cheirality votes run on the GPU
    (csrc/bootstrap.hip); the host keeps the sequential accept rule and the bookkeeping classes."""
    camera = sequence.get_camera()
    triangulator, pose_estimator = make_estimators(camera, ransac_threshold)
    init_frame = next(sequence)
    state = State(init_frame)
    next(sequence)                                                       # frame 1 is skipped
    new_frame = next(sequence)
    if tracker_setup is not None:
        tracker_setup()
    tracker = Tracker(init_frame, mode=tracker_mode)
    matches = tracker.trackFeatures(state.curr_frame, new_frame)
    state.update_from_matches(matches)
    t0 = time.perf_counter()
    M, landmarks, inliers = triangulator.triangulate_matches(matches)
    state.bootstrap_info = {"relative_pose_seconds": time.perf_counter() - t0, "correspondences": int(len(inliers)),
                            "inliers": int(np.sum(inliers))}
    f2 = matches.frame2.features
    outliers = np.zeros(shape=(f2.length,), dtype=bool)
    outliers[f2.match_inliers] = ~inliers
    state.update_with_local_pose(M)
    inliers_mask = np.zeros_like(f2.matched_candidate_inliers).astype(bool)
    inliers_mask[f2.matched_candidate_inliers] = inliers
    state.update_with_local_landmarks(landmarks[inliers], inliers_mask)
    state.reset_outliers(outliers)
    return state, tracker, triangulator, pose_estimator


def run(sequence: Sequence, tracker_mode: str = "klt", max_frames: int = None, verbose: bool = False):
    """The reference's loop through the drop-in classes, one call per stage (host bookkeeping, host <-> device
    copies around every kernel).  Returns dict(trajectory (n, 4, 4) camera-to-world, n_landmarks, frame_seconds)."""
    state, tracker, triangulator, pose_estimator = bootstrap(sequence, tracker_mode)
    trajectory = [np.eye(4), state.get_pose()]
    n_landmarks = [len(state.curr_frame.features.triangulated_inliers_landmarks)]
    seconds = []

    # ---- steady state ----
    for k, new_frame in enumerate(sequence):
        if max_frames is not None and k >= max_frames:
            break
        t0 = time.perf_counter()
        matches = tracker.trackFeatures(state.curr_frame, new_frame)
        f2 = matches.frame2.features
        (rmatrix, tvec), inliers = pose_estimator.estimate_pose(
            Features(keypoints=f2.triangulated_inliers_keypoints, landmarks=f2.triangulated_inliers_landmarks))
        outliers = np.zeros(shape=(f2.length,), dtype=bool)
        outliers[f2.triangulate_inliers] = ~inliers
        state.update_from_matches(matches)
        state.update_with_world_pose(np.concatenate((rmatrix, tvec), axis=1))
        state.reset_outliers(outliers)
        state.compute_candidates()
        feats = state.curr_frame.features
        assert np.sum(feats.candidate_mask) <= np.sum(feats.matched_candidate_inliers)
        if np.sum(feats.candidate_mask) > 0:
            world = triangulator.triangulate_candidates(feats, current_pose=state.get_pose())
            state.update_with_world_landmarks(world, matches.frame2.features.candidate_mask)
        seconds.append(time.perf_counter() - t0)
        trajectory.append(state.get_pose())
        n_landmarks.append(len(feats.triangulated_inliers_landmarks))
        if verbose:
            print("frame %3d: %4d keypoints, %4d landmarks, %.1f ms" % (k + 3, feats.length, n_landmarks[-1],
                                                                         seconds[-1] * 1e3))
    return dict(trajectory=np.array(trajectory), n_landmarks=np.array(n_landmarks), frame_seconds=np.array(seconds))


def _gray(image):
    from vo.features.klt import _gray as g
    return g(image)


def run_on_device(sequence: Sequence, max_frames: int = None, n_keypoints: int = 2000, klt_win: int = 17,
                  klt_max_level: int = 2, hyp: int = 4000, context=None, verbose: bool = False,
                  redetect_start_pose: str = "current", bootstrap_win: int = None, bootstrap_max_level: int = None,
                  bootstrap_threshold: float = 0.25):
    """Same loop, same bootstrap, but the steady state runs as the device-resident pipeline (vo_pipeline_*):
    after the host bootstrap the Features / State arrays are handed to the GPU once, every later frame costs one
    image upload and one call, and nothing but the pose record comes back.  KLT tracker mode with the Harris
    detector (BASELINE.json configs[1]); P3P-RANSAC as main.py:194-201 configures it (1.25 px, confidence 0.9999)
    with `hyp` hypotheses solved and scored per launch (a frame whose sequential rule needs more -- main.py allows
    10000 iterations -- gets further launches of hypotheses until the rule is done; the loop's state stays on the
    device).  redetect_start_pose: "identity" is the reference's
    update_features (klt.py:148-153: re-detected keypoints start their track at np.eye(4), so away from the origin
    they triangulate against a wrong baseline and can take the estimate with them); "current" starts them at the
    pose of the frame they were found on."""
    from vo import _native
    from vo.features.klt import KLTTracker
    ctx = context or _native.default_context()
    saved = (dict(KLTTracker._feature_params), dict(KLTTracker._lk_params))

    def setup():
        # the bootstrap tracks Shi-Tomasi corners (the reference's find_corners, klt.py:98), as many as the
        # pipeline's detector keeps per frame
        KLTTracker._feature_params = dict(saved[0], maxCorners=n_keypoints)
        # (bootstrap_*: the two bootstrap frames are further apart than consecutive ones; on large frames the loop's own
        #  window and the reference's 0.25 px epipolar threshold can settle on a wrong model, bench.py: bootstrap_state)
        bw = bootstrap_win or klt_win
        KLTTracker._lk_params = dict(saved[1], winSize=(bw, bw),
                                     maxLevel=klt_max_level if bootstrap_max_level is None else bootstrap_max_level)

    try:
        state, tracker, _, _ = bootstrap(sequence, "klt", tracker_setup=setup, ransac_threshold=bootstrap_threshold)
    finally:
        KLTTracker._feature_params, KLTTracker._lk_params = saved
    frame = state.curr_frame
    img = _gray(frame.image)
    H, W = img.shape
    K = np.asarray(sequence.get_camera().intrinsic_matrix, np.float64)
    SLOTS = 4
    pipe = _native.Pipeline(ctx, H, W, SLOTS, K, n_keypoints=n_keypoints, klt_win=klt_win, klt_max_level=klt_max_level,
                            hyp=hyp, p3p_threshold=1.25 ** 2, outlier_ratio=0.9, confidence=0.9999, max_iterations=10000,
                            refine_iters=20, bearing_threshold=state._bearing_threshold,
                            redetect_start_pose=redetect_start_pose)
    pipe.set_frame(0, img)
    pipe.set_state(0, frame.features, state.curr_pose, state.prev_pose, num_features=tracker._tracker._num_features)
    trajectory = [np.eye(4), state.get_pose()]
    n_landmarks = [len(frame.features.triangulated_inliers_landmarks)]
    seconds, results = [], []
    # Frames go through a ring of pinned buffers (the grey conversion writes into them) and are uploaded on the
    # pipeline's upload stream ONE STEP AHEAD of their use (vo_pipeline_set_frame_pinned): while step k -> k+1 runs,
    # frame k+2 -- read ahead from the sequence, as a file or dataset reader can -- crosses PCIe beside it.  One frame
    # of look-ahead on the results as before: the pose of frame k is read back after frame k+1 has been submitted.
    ring = [ctx.pinned_empty((H, W)) for _ in range(SLOTS)]
    frames = iter(sequence)
    taken = 0

    def take():
        nonlocal taken
        if max_frames is not None and taken >= max_frames:
            return None
        f = next(frames, None)
        if f is not None:
            taken += 1
        return f

    def put(s, f):
        ring[s][...] = _gray(f.image)
        pipe.set_frame(s, ring[s], pinned=True)

    slot, pending = 0, 0
    ahead = take()
    if ahead is not None:
        put(1, ahead)
    while ahead is not None:
        t0 = time.perf_counter()
        nxt = (slot + 1) % SLOTS
        if pending == 2:
            results.append(pipe.collect())
            pending -= 1
        t1 = time.perf_counter()
        ahead = take()                                   # the frame of the NEXT step: its slot was read last by a collected step
        t2 = time.perf_counter()                         # (reading / decoding / rendering the frame is the sequence's time, not the loop's)
        if ahead is not None:
            put((slot + 2) % SLOTS, ahead)
        pipe.submit(slot, nxt)
        if ahead is not None:
            pipe.prepare((slot + 2) % SLOTS)            # its pyramid too, behind this step's tracker
        pending += 1
        slot = nxt
        seconds.append(time.perf_counter() - t0 - (t2 - t1))
    while pending:
        results.append(pipe.collect())
        pending -= 1
    for r in results:
        trajectory.append(r.pose_world_cam())
        n_landmarks.append(r.n_landmarks)
        if verbose:
            print("%4d in, %4d tracked, %4d landmarks, %4d inliers, %3d candidates%s" % (
                r.n_features_in, r.n_tracked, r.n_landmarks, r.n_inliers, r.n_candidates,
                ", re-detected" if r.redetected else ""))
    features = pipe.get_features()
    pipe.close()
    return dict(trajectory=np.array(trajectory), n_landmarks=np.array(n_landmarks), frame_seconds=np.array(seconds),
                results=results, features=features)


def trajectory_error(result, sequence: Sequence):
    """RMS position error against the analytic ground truth after fitting the one free scale
    of the monocular bootstrap.  Trajectory index 0 is frame 0, index i >= 1 is frame i + 1."""
    traj = result["trajectory"]
    frames = [0] + list(range(2, 2 + len(traj) - 1))
    gt = np.stack([np.linalg.inv(sequence.ground_truth_pose(0)) @ sequence.ground_truth_pose(f) for f in frames])
    p, q = traj[:, :3, 3], gt[:, :3, 3]
    scale = float(np.sum(p * q) / max(np.sum(p * p), 1e-30))
    return dict(scale=scale, rms=float(np.sqrt(np.mean(np.sum((scale * p - q) ** 2, axis=1)))),
                path_length=float(np.sum(np.linalg.norm(np.diff(q, axis=0), axis=1))))


if __name__ == "__main__":
    seq = Sequence("synthetic", n_frames=30, height=480, width=640, channels=3)
    out = run(seq, "klt", verbose=True)
    print(trajectory_error(out, seq), "mean ms/frame", 1e3 * out["frame_seconds"].mean())
    seq = Sequence("synthetic", n_frames=30, height=480, width=640, channels=3)
    out = run_on_device(seq, n_keypoints=500, verbose=True)
    print(trajectory_error(out, seq), "mean ms/frame", 1e3 * out["frame_seconds"].mean())
