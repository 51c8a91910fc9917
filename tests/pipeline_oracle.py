"""CPU oracle of the device-resident frame loop (vo_pipeline_*): the reference driver's steady state
(src/main.py:248-286, KLT tracker mode with the Harris detector of BASELINE.json configs[1]) composed from

  * the bookkeeping classes of the `vo` package (Features / Frame / Matches / State -- host NumPy, pinned
    bit for bit to the reference's own classes by tests/golden/bookkeeping.npz) and its KLTTracker shell
    (the reference's klt.py:191-280 call sequence), and
  * the CPU oracles for every piece of arithmetic: oracle/harris_np (pinned to the reference), oracle/csrc
    klt.c / p3p.c (parity unpinned vs OpenCV), oracle/ransac_np (the reference's loop, pinned),
    oracle/refine_np (pinned to SciPy), oracle/dlt_np (pinned).

TEST INFRASTRUCTURE ONLY: nothing here touches the GPU."""
import numpy as np

from oracle import dlt_np, harris_np, native, ransac_np, refine_np


class OracleContext:
    """Stands in for vo._native.Context inside vo.features.klt.KLTTracker: same method names, CPU oracles."""

    def harris_keypoints(self, img, patch, kappa, n, r):
        return harris_np.nms_keypoints_fast(harris_np.harris_scores(img, patch, kappa), n, r)[:, :, 0]

    def klt_track(self, prev, nxt, prev_xy, win=17, max_level=2, max_iter=10, eps=0.03, min_eig=1e-4):
        return native.klt_track(prev, nxt, np.asarray(prev_xy, np.float32).reshape(-1, 2), win=win, max_level=max_level,
                                max_iter=max_iter, eps=eps, min_eig=min_eig)


def make_tracker(frame, n_keypoints, win, max_level, redetect_start_pose="identity"):
    """KLTTracker (the product's host shell of klt.py) wired to the oracle context.  redetect_start_pose =
    "current": re-detected keypoints start their track at `current_pose` (set by the loop before every call)
    instead of the reference's np.eye(4) (klt.py:148-153) -- the pipeline's redetect_start_pose = 1."""
    from vo.features.klt import KLTTracker, TERM_CRITERIA_COUNT, TERM_CRITERIA_EPS

    class OracleKLT(KLTTracker):
        _detector = "harris"
        _harris_params = dict(patch_size=9, kappa=0.09, num_keypoints=n_keypoints, nonmaximum_supression_radius=5)
        _lk_params = dict(winSize=(win, win), maxLevel=max_level,
                          criteria=(TERM_CRITERIA_EPS | TERM_CRITERIA_COUNT, 10, 0.03))
        current_pose = None

        def update_features(self, new_keypoints):
            feats = super().update_features(new_keypoints)
            if redetect_start_pose == "current" and len(new_keypoints) > 0:
                poses = feats.poses.copy()
                poses[-len(new_keypoints):] = self.current_pose
                feats.poses = poses
            return feats

    return OracleKLT(frame, context=OracleContext())


class OracleLoop:
    def __init__(self, stream, n_keypoints, win, max_level, p3p_threshold=1.0, outlier_ratio=0.9, confidence=0.99,
                 max_iterations=1000, refine_iters=20, bearing_threshold=0.0075, seed=2023,
                 redetect_start_pose="identity", tracker="klt", match_ratio=0.8):
        """tracker = "sift": src/vo/features/tracker.py:60-61 -> sift.py:23-56 (detect + describe the new frame, the
        n_keypoints strongest; 2-NN + ratio + uniqueness against the current frame's descriptors; Matches from the pairs)
        instead of the KLT tracker."""
        self.tracker_mode, self.match_ratio = tracker, match_ratio
        from vo.primitives import Frame
        from vo.sensors import Camera
        self.stream = stream
        self.K = stream.K
        self.cam = Camera(intrinsic_matrix=self.K)
        self.Frame = Frame
        self.cfg = dict(N=n_keypoints, win=win, lvl=max_level, refine=refine_iters, bearing=bearing_threshold,
                        redetect=redetect_start_pose)
        self.rs = ransac_np.Ransac(4, np.arange(4), None, None, p3p_threshold, outlier_ratio, confidence, max_iterations,
                                   adaptive=True, p3p=True)
        self.rs.rng = np.random.default_rng(seed)
        self.tracker = None
        self.state = None

    def frame(self, idx, features=None):
        return self.Frame(self.stream.image(idx), features=features, sensor=self.cam, intrinsics=self.K)

    def set_state(self, idx, features, curr_pose, prev_pose):
        """`features`: vo.primitives.Features of frame idx (copied)."""
        import copy
        from vo.primitives import State
        f = self.frame(idx, copy.deepcopy(features))
        self.state = State(f, bearing_threshold=self.cfg["bearing"])
        self.state.curr_pose = np.array(curr_pose, np.float64)
        self.state.prev_pose = np.array(prev_pose, np.float64)
        self.state.prev_frame = f
        if self.tracker_mode in ("sift", "harris"):
            return
        # the tracker shell: built around the frame without touching its features
        keep = f.features
        self.tracker = make_tracker(self.frame(idx), self.cfg["N"], self.cfg["win"], self.cfg["lvl"], self.cfg["redetect"])
        self.tracker._num_features = self.cfg["N"]
        f.features = keep

    def step(self, next_idx):
        st, K = self.state, self.K
        new = self.frame(next_idx)
        n_before = st.curr_frame.features.length
        if self.tracker_mode == "sift":
            from vo.primitives import Features, Matches
            kp2, desc2 = native.sift(self.stream.image(next_idx), cap=self.cfg["N"])
            new.features = Features(kp2[:, :2].astype(np.float64).reshape(-1, 2, 1))
            new.features.descriptors = desc2
            good = native.match_knn2_ratio(st.curr_frame.features.descriptors, desc2, self.match_ratio)[0]
            if len(good) == 0:
                good = np.empty((0, 2), dtype=int)
            matches = Matches(st.curr_frame, new, matches=good)
            self.n_new, self.n_pairs = len(kp2), len(good)
        elif self.tracker_mode == "harris":
            # harris.py:50-84: extractKeypoints + extractDescriptors on the new frame, matchDescriptor (ratio 0.85)
            from vo.primitives import Features, Matches
            img = self.stream.image(next_idx)
            kp2 = harris_np.nms_keypoints_fast(harris_np.harris_scores(img, 9, 0.09), self.cfg["N"], 5)
            new.features = Features(kp2.astype(np.float64))
            new.features.descriptors = harris_np.patch_descriptors(img, kp2, 9)
            d1 = st.curr_frame.features.descriptors.astype(np.float32)
            d2 = new.features.descriptors.astype(np.float32)
            good = native.match_knn2_ratio(d1.reshape(len(d1), -1), d2.reshape(len(d2), -1), 0.85)[0]
            if len(good) == 0:
                good = np.empty((0, 2), dtype=int)
            matches = Matches(st.curr_frame, new, matches=good)
            self.n_new, self.n_pairs = len(kp2), len(good)
        else:
            self.tracker.current_pose = st.get_pose().copy()
            matches = self.tracker.track_features(st.curr_frame, new)
        f2 = matches.frame2.features
        X = np.ascontiguousarray(f2.triangulated_inliers_landmarks[:, :, 0], np.float64)
        x = np.ascontiguousarray(f2.triangulated_inliers_keypoints[:, :, 0], np.float64)
        rs = self.rs
        rs.model_fn = lambda idx: native.p3p_solve(X[np.asarray(idx).reshape(-1)], x[np.asarray(idx).reshape(-1)], K)
        rs.error_fn = lambda m, pop: native.reproj_errors(X, x, K, m[0], m[1])
        n0 = len(rs.trace)
        (R, t), inl = rs.find_best_model(np.arange(len(X)))
        Rr, tr, it, cost = R, t, -1, 0.0
        if self.cfg["refine"] > 0:
            Rr, tr, it, cost = refine_np.refine_pose(X[inl], x[inl], K, R, t, max_iter=self.cfg["refine"])
        outliers = np.zeros(f2.length, dtype=bool)
        outliers[f2.triangulate_inliers] = ~inl
        st.update_from_matches(matches)
        st.update_with_world_pose(np.concatenate((Rr, np.asarray(tr).reshape(3, 1)), axis=1))
        st.reset_outliers(outliers)
        st.compute_candidates()
        feats = st.curr_frame.features
        cand = feats.candidate_mask.copy()
        n_cand = int(cand.sum())
        n_state2_before = int((feats.state == 2).sum())
        if n_cand > 0:
            P1, P2 = dlt_np.candidate_projections(K, feats.poses[cand], st.get_pose())
            Xw = dlt_np.linear_triangulation(feats.tracks[cand][:, :, 0], feats.keypoints[cand][:, :, 0], P1, P2)
            st.update_with_world_landmarks(Xw.reshape(-1, 3, 1), cand)
        return dict(n_before=n_before,
                    n_tracked=f2.length, n_tri=len(X), R=R, t=np.asarray(t).reshape(3), R_ref=Rr,
                    t_ref=np.asarray(tr).reshape(3), refine_iters=it, inliers=inl, n_inliers=int(inl.sum()),
                    draws=len(rs.trace) - n0, iters=rs.iterations_done, n_cand=n_cand, candidate_mask=cand,
                    n_landmarks=int((feats.state == 2).sum()), n_state2_before=n_state2_before,
                    features=feats, pose=st.get_pose().copy(), n_iterations=rs.n_iterations,
                    outlier_ratio=rs.outlier_ratio)


def initial_features(stream, idx, n_keypoints, depth_scale=1.0):
    """A deterministic starting state for frame idx without the two-view bootstrap: the Harris keypoints of the
    frame, the first two thirds triangulated (landmarks from the stream's analytic depth, expressed in the
    world frame of the analytic pose), the rest matched tracks that started here.  Exercises all three
    states from the first step on."""
    from vo.primitives import Features
    K = stream.K
    kp = harris_np.nms_keypoints_fast(harris_np.harris_scores(stream.image(idx), 9, 0.09), n_keypoints, 5)
    kp = kp.astype(np.float32)
    f = Features(keypoints=kp)
    n = f.length
    T = stream.T_world_cam(idx)
    z = stream.depth(idx)[kp[:, 1, 0].astype(int), kp[:, 0, 0].astype(int)].astype(np.float64) * depth_scale
    xc = (kp[:, 0, 0].astype(np.float64) - K[0, 2]) / K[0, 0] * z
    yc = (kp[:, 1, 0].astype(np.float64) - K[1, 2]) / K[1, 1] * z
    land = np.stack([T[r, 0] * xc + T[r, 1] * yc + T[r, 2] * z + T[r, 3] for r in range(3)], axis=1)
    n_tri = (2 * n) // 3
    # group order as Matches leaves it: triangulated first, then matched
    f.state = np.concatenate([2 * np.ones(n_tri), np.ones(n - n_tri)])
    f.landmarks[:n_tri] = land[:n_tri].reshape(-1, 3, 1)
    f.tracks = np.concatenate([np.full((n_tri, 2, 1), np.nan), kp[n_tri:].astype(np.float64)])
    f.poses = np.concatenate([np.full((n_tri, 4, 4), np.nan), np.stack([T] * (n - n_tri))])
    return f, T


def initial_sift_features(stream, idx, n_keypoints):
    """Starting state for the SIFT tracker mode: the frame's SIFT keypoints (the n strongest) with their descriptors,
    the first two thirds triangulated from the stream's analytic depth along the keypoint's own ray, the rest matched
    tracks that started here."""
    from vo.primitives import Features
    K = stream.K
    kp6, desc = native.sift(stream.image(idx), cap=n_keypoints)
    kp = kp6[:, :2].astype(np.float64).reshape(-1, 2, 1)
    f = Features(keypoints=kp)
    f.descriptors = desc
    n = f.length
    T = stream.T_world_cam(idx)
    H, W = stream.image(idx).shape
    yi = np.clip(np.rint(kp[:, 1, 0]).astype(int), 0, H - 1)
    xi = np.clip(np.rint(kp[:, 0, 0]).astype(int), 0, W - 1)
    z = stream.depth(idx)[yi, xi].astype(np.float64)
    xc = (kp[:, 0, 0] - K[0, 2]) / K[0, 0] * z
    yc = (kp[:, 1, 0] - K[1, 2]) / K[1, 1] * z
    land = np.stack([T[r, 0] * xc + T[r, 1] * yc + T[r, 2] * z + T[r, 3] for r in range(3)], axis=1)
    n_tri = (2 * n) // 3
    f.state = np.concatenate([2 * np.ones(n_tri), np.ones(n - n_tri)])
    f.landmarks[:n_tri] = land[:n_tri].reshape(-1, 3, 1)
    f.tracks = np.concatenate([np.full((n_tri, 2, 1), np.nan), kp[n_tri:]])
    f.poses = np.concatenate([np.full((n_tri, 4, 4), np.nan), np.stack([T] * (n - n_tri))])
    return f, T


def initial_harris_features(stream, idx, n_keypoints):
    """Starting state for the Harris tracker mode: initial_features' keypoints and landmarks with their raw-patch
    descriptors (harris.py:160-194)."""
    f, T = initial_features(stream, idx, n_keypoints)
    f.descriptors = harris_np.patch_descriptors(stream.image(idx), f.keypoints, 9)
    return f, T
