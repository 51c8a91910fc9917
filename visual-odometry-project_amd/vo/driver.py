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


def run(sequence: Sequence, tracker_mode: str = "klt", max_frames: int = None, verbose: bool = False):
    """Returns dict(trajectory (n, 4, 4) camera-to-world, n_landmarks, frame_seconds)."""
    camera = sequence.get_camera()
    triangulator = LandmarksTriangulator(camera1=camera, camera2=camera, use_ransac=True, use_opencv=True,
                                         outlier_ratio=0.9, ransac_threshold=0.25, ransac_confidence=0.999)
    pose_estimator = P3PPoseEstimator(use_opencv=True, intrinsic_matrix=camera.intrinsic_matrix,
                                      inlier_threshold=1.25, outlier_ratio=0.9, confidence=0.9999,
                                      nonlinear_refinement=True)
    # ---- bootstrap ----
    init_frame = next(sequence)
    state = State(init_frame)
    next(sequence)                                                       # frame 1 is skipped
    new_frame = next(sequence)
    tracker = Tracker(init_frame, mode=tracker_mode)
    matches = tracker.trackFeatures(state.curr_frame, new_frame)
    state.update_from_matches(matches)
    M, landmarks, inliers = triangulator.triangulate_matches(matches)
    f2 = matches.frame2.features
    outliers = np.zeros(shape=(f2.length,), dtype=bool)
    outliers[f2.match_inliers] = ~inliers
    state.update_with_local_pose(M)
    inliers_mask = np.zeros_like(f2.matched_candidate_inliers).astype(bool)
    inliers_mask[f2.matched_candidate_inliers] = inliers
    state.update_with_local_landmarks(landmarks[inliers], inliers_mask)
    state.reset_outliers(outliers)
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
