"""Device-resident frame loop (vo_pipeline_*, include/vo_hip.h): the steady state of the reference
driver (src/main.py:248-286, KLT tracker mode) with the Features / State / RANSAC bookkeeping kept in
HBM.  The pipeline takes images only; `set_state` hands over what the bootstrap produced and
`get_state` returns the reference's Features arrays of the current frame.  With `sequences=S` the pipeline
advances S independent streams per launch (`seq=` addresses one of them, `collect_all` returns all records)."""
import ctypes as C

import numpy as np


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _as_4x4(pose):
    pose = np.asarray(pose, np.float64)
    if pose.shape == (3, 4):
        pose = np.vstack([pose, [0.0, 0.0, 0.0, 1.0]])
    return np.ascontiguousarray(pose)


class Pipeline:
    def __init__(self, ctx, H, W, n_frames, K, n_keypoints=2000, harris_patch=9, harris_kappa=0.09, nms_radius=5,
                 klt_win=15, klt_max_level=2, klt_max_iter=10, klt_eps=0.03, klt_min_eig=1e-4,
                 klt_err_threshold=100.0, hyp=1000, p3p_threshold=1.0, outlier_ratio=0.9, confidence=0.99,
                 max_iterations=1000, seed=2023, refine_iters=0, feature_cap=0, bearing_threshold=0.0075,
                 redetect_fraction=0.8, debug_fault_every=0, redetect_start_pose="identity", sequences=1,
                 detect_margin=0.01, debug_never_detect=0, detect_losses=2.5, tracker="klt", sift_cap=0, match_ratio=0.0):
        from vo import _native
        self.ctx = ctx
        self.cfg = _native.PipelineConfig()
        c = self.cfg
        c.H, c.W, c.n_frames = H, W, n_frames
        c.n_keypoints, c.harris_patch, c.nms_radius, c.harris_kappa = n_keypoints, harris_patch, nms_radius, harris_kappa
        c.klt_win, c.klt_max_level, c.klt_max_iter, c.hyp = klt_win, klt_max_level, klt_max_iter, hyp
        c.klt_eps, c.klt_min_eig, c.klt_err_threshold = klt_eps, klt_min_eig, klt_err_threshold
        c.p3p_thr_sq = p3p_threshold
        c.ransac_outlier_ratio, c.ransac_confidence = outlier_ratio, confidence
        c.ransac_max_iterations = -1 if max_iterations is None or max_iterations == np.inf else int(max_iterations)
        c.refine_iters = int(refine_iters)
        c.feature_cap = int(feature_cap)
        c.bearing_threshold = float(bearing_threshold)
        c.redetect_fraction = float(redetect_fraction)
        c.debug_fault_every = int(debug_fault_every)
        c.redetect_start_pose = {"identity": 0, "current": 1}[redetect_start_pose]
        c.sequences = int(sequences)
        c.detect_margin = float(detect_margin)       # < 0: the detector runs on every frame
        c.debug_never_detect = int(debug_never_detect)
        c.detect_losses = float(detect_losses)
        c.tracker_mode = {"klt": 0, "sift": 1, "harris": 2}[tracker]         # src/vo/features/tracker.py:54-63
        c.sift_cap = int(sift_cap)
        c.match_ratio = float(match_ratio)
        self.tracker = tracker
        self.sequences = int(sequences)
        K = np.asarray(K, np.float64).reshape(3, 3)
        self.K = K
        for i, v in enumerate(K.reshape(9)):
            c.K[i] = v
        # the reference normalises keypoints with np.linalg.inv(K) (src/vo/sensors/camera.py:88)
        for i, v in enumerate(np.linalg.inv(K).reshape(9)):
            c.Kinv[i] = v
        h = C.c_void_p()
        ctx._chk(ctx._lib.vo_pipeline_create(ctx._h, C.byref(c), C.byref(h)))
        self._h = h
        self._pinned_src = {}
        ctx._pipelines.add(self)
        self.cap = ctx._lib.vo_pipeline_feature_cap(self._h)
        self.seed(np.random.default_rng(seed))

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._lib.vo_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs ----
    def set_frame(self, idx, img, seq=0, pinned=None):
        """Frame slot idx <- img.  pinned: img lies in Context.pinned_empty memory and is uploaded from there (no staging
        copy, DMA beside the kernels; img must stay unchanged until frame_uploaded(idx) or the collect of a step that
        read the slot); None: decided by where img lies."""
        img = _c(img, np.uint8)
        assert img.shape == (self.cfg.H, self.cfg.W)
        if pinned is None:
            pinned = self.ctx.is_pinned(img)
        if pinned:
            self._pinned_src[(int(seq), int(idx))] = img          # (kept alive while the DMA may read it)
            self.ctx._chk(self.ctx._lib.vo_pipeline_set_frame_pinned(self._h, int(seq), int(idx), _ptr(img)))
        else:
            self.ctx._chk(self.ctx._lib.vo_pipeline_set_frame_seq(self._h, int(seq), int(idx), _ptr(img)))

    def prepare(self, idx):
        """Hint: frame slot idx is the `next` of the coming submit -- its pyramid is built now, off that step's critical
        path (vo_pipeline_prepare).  Results do not depend on it."""
        self.ctx._chk(self.ctx._lib.vo_pipeline_prepare(self._h, int(idx)))

    def frame_uploaded(self, idx, wait=False):
        rc = self.ctx._lib.vo_pipeline_frame_uploaded(self._h, int(idx), 1 if wait else 0)
        if rc < 0:
            self.ctx._chk(rc)
        return bool(rc)

    def seed(self, generator):
        """The estimator's generator (RANSAC.rng, src/vo/algorithms/ransac.py:52)."""
        from vo import _native
        pcg = _native.Pcg64.from_generator(generator)
        self.ctx._chk(self.ctx._lib.vo_pipeline_seed(self._h, C.byref(pcg)))

    def rng_state_into(self, generator, seq=0):
        """Writes the estimator generator's state after the last collected step into `generator`."""
        from vo import _native
        pcg = _native.Pcg64()
        self.ctx._chk(self.ctx._lib.vo_pipeline_get_rng_seq(self._h, int(seq), C.byref(pcg)))
        pcg.to_generator(generator)

    def set_state(self, idx, features, curr_pose, prev_pose=None, num_features=None, seq=0):
        """Hands over `features` (a vo.primitives.Features: the current frame's, e.g. after the bootstrap) and
        State's poses (4x4 camera-to-world) for frame slot `idx`."""
        n = features.length
        kp = _c(np.asarray(features.keypoints).reshape(n, 2), np.float64)
        state = _c(np.asarray(features.state).reshape(n), np.uint8)
        land = _c(np.asarray(features.landmarks).reshape(n, 3), np.float64)
        tracks = _c(np.asarray(features.tracks).reshape(n, 2), np.float64)
        poses = _c(np.asarray(features.poses).reshape(n, 16), np.float64)
        T_wc = _as_4x4(curr_pose)
        T_wc_prev = _as_4x4(prev_pose if prev_pose is not None else curr_pose)
        T_cw, T_cw_prev = _c(np.linalg.inv(T_wc), np.float64), _c(np.linalg.inv(T_wc_prev), np.float64)
        nf = int(num_features if num_features is not None else self.cfg.n_keypoints)
        self.ctx._chk(self.ctx._lib.vo_pipeline_set_state_seq(self._h, int(seq), int(idx), n, _ptr(kp), _ptr(state),
                                                              _ptr(land), _ptr(tracks), _ptr(poses), _ptr(T_wc),
                                                              _ptr(T_cw), _ptr(T_wc_prev), _ptr(T_cw_prev), nf))
        if self.tracker in ("sift", "harris"):
            desc = _c(np.asarray(features.descriptors).reshape(n, 128 if self.tracker == "sift" else 361), np.float32)
            self.ctx._chk(self.ctx._lib.vo_pipeline_set_descriptors(self._h, _ptr(desc), n))

    def checkpoint(self):
        """Keeps a copy of every sequence's Features / State as they are now (nothing in flight) in HBM."""
        self.ctx._chk(self.ctx._lib.vo_pipeline_checkpoint(self._h))

    def rewind(self):
        """Puts the checkpoint back (asynchronously, nothing in flight): the next submit starts from its frame again.
        The estimator's RANSAC fields and generator go on, as they would on the reference's estimator object."""
        self.ctx._chk(self.ctx._lib.vo_pipeline_rewind(self._h))

    # ---- outputs ----
    def get_state(self, seq=0):
        """dict with the reference's Features arrays of the current frame (shapes as in
        src/vo/primitives/features.py) plus curr_pose / prev_pose / RANSAC fields."""
        from vo import _native
        cap = self.cap
        n = C.c_int32()
        nf = C.c_int32()
        kp = np.empty((cap, 2), np.float64)
        state = np.empty(cap, np.uint8)
        cand = np.empty(cap, np.uint8)
        land = np.empty((cap, 3), np.float64)
        tracks = np.empty((cap, 2), np.float64)
        poses = np.empty((cap, 4, 4), np.float64)
        T, Tp = np.empty((4, 4)), np.empty((4, 4))
        rs = _native.RansacState()
        self.ctx._chk(self.ctx._lib.vo_pipeline_get_state_seq(self._h, int(seq), C.byref(n), _ptr(kp), _ptr(state),
                                                              _ptr(cand), _ptr(land), _ptr(tracks), _ptr(poses),
                                                              _ptr(T), _ptr(Tp), C.byref(rs), C.byref(nf)))
        n = n.value
        return dict(n=n, keypoints=kp[:n].reshape(n, 2, 1).copy(), state=state[:n].astype(np.float64),
                    candidate_mask=cand[:n].astype(bool), landmarks=land[:n].reshape(n, 3, 1).copy(),
                    tracks=tracks[:n].reshape(n, 2, 1).copy(), poses=poses[:n].copy(), curr_pose=T, prev_pose=Tp,
                    n_iterations=int(rs.n_iterations), outlier_ratio=float(rs.outlier_ratio),
                    num_features=nf.value)

    def get_features(self, seq=0):
        """The current frame's features as a vo.primitives.Features object."""
        from vo.primitives import Features
        s = self.get_state(seq)
        f = Features(keypoints=s["keypoints"], landmarks=s["landmarks"])
        f.state, f.tracks, f.poses, f.candidate_mask = s["state"], s["tracks"], s["poses"], s["candidate_mask"]
        return f

    def get_detection(self):
        kp = np.empty((self.cfg.n_keypoints, 2), np.float64)
        self.ctx._chk(self.ctx._lib.vo_pipeline_get_detection(self._h, _ptr(kp)))
        return kp

    # ---- frames ----
    def step(self, prev_idx, next_idx):
        from vo import _native
        r = _native.StepResult()
        self.ctx._chk(self.ctx._lib.vo_pipeline_step(self._h, int(prev_idx), int(next_idx), C.byref(r)))
        return r

    def submit(self, prev_idx, next_idx):
        """Enqueue the frame's GPU work and return (at most two steps in flight)."""
        self.ctx._chk(self.ctx._lib.vo_pipeline_submit(self._h, int(prev_idx), int(next_idx)))

    def collect(self):
        """Wait for the oldest submitted frame's result record."""
        from vo import _native
        r = _native.StepResult()
        self.ctx._chk(self.ctx._lib.vo_pipeline_collect(self._h, C.byref(r)))
        return r

    def collect_all(self):
        """Wait for the oldest submitted frame's records of all sequences (a list of StepResult)."""
        from vo import _native
        rs = (_native.StepResult * self.sequences)()
        self.ctx._chk(self.ctx._lib.vo_pipeline_collect_all(self._h, rs))
        return list(rs)

    def bookkeeping(self, phases, new_keypoints=None, pairs=None, pose_world_cam=None, p3p_inliers=None):
        """One frame's bookkeeping with the estimators' outputs given by the caller (vo_pipeline_bookkeeping)."""
        if phases & 1:
            kp = _c(np.asarray(new_keypoints).reshape(-1, 2), np.float64)
            pr = _c(np.asarray(pairs).reshape(-1, 2), np.int32)
            T_wc = _as_4x4(pose_world_cam)
            T_cw = _c(np.linalg.inv(T_wc), np.float64)
            inl = None if p3p_inliers is None else _c(np.asarray(p3p_inliers).reshape(-1), np.uint8)
            self.ctx._chk(self.ctx._lib.vo_pipeline_bookkeeping(self._h, int(phases), _ptr(kp), kp.shape[0], _ptr(pr),
                                                                pr.shape[0], _ptr(T_wc), _ptr(T_cw), _ptr(inl)))
        else:
            self.ctx._chk(self.ctx._lib.vo_pipeline_bookkeeping(self._h, int(phases), None, 0, None, 0, None, None, None))

    def ransac_bound(self, outlier_ratio):
        return int(self.ctx._lib.vo_pipeline_ransac_bound(self._h, float(outlier_ratio)))

    # ---- profiling / shared map ----
    def prof_read(self, kernel_id):
        ms, n = C.c_double(), C.c_int64()
        self.ctx._chk(self.ctx._lib.vo_pipeline_prof_read(self._h, int(kernel_id), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def prof_reset(self):
        self.ctx._chk(self.ctx._lib.vo_pipeline_prof_reset(self._h))

    def export_state_post(self, result, cap, d_record, seq=0):
        """Queues the shared-map record of the last collected step on the pipeline's stream (no synchronisation)."""
        self.ctx._chk(self.ctx._lib.vo_pipeline_export_state_post_seq(self._h, int(seq), C.byref(result), int(cap),
                                                                      C.c_void_p(d_record)))

    def export_state_join(self, consumer_stream=None):
        """Orders the records posted so far before later work of `consumer_stream`, and later records after
        what that stream holds now."""
        self.ctx._chk(self.ctx._lib.vo_pipeline_export_state_join(self._h, C.c_void_p(consumer_stream or 0)))
