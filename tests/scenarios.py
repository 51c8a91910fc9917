"""Scripted call sequences shared by tools/make_golden.py (run against the
reference's classes, in the build container only) and the parity tests (run
against this repo's classes).  Inputs are seeded; every array a step produces is
returned so the golden pins the bookkeeping bit-for-bit.

The call order follows the reference driver src/main.py:204-230 (bootstrap) and
:248-286 (steady state), with ground-truth poses standing in for the two OpenCV
calls the reference makes there (findFundamentalMat / solvePnP).
"""
import numpy as np


def synthetic_image(h, w, seed, block=8, noise=4.0):
    """Seeded block texture + noise, uint8 (h, w)."""
    rng = np.random.default_rng(seed)
    gh, gw = -(-h // block), -(-w // block)
    cells = rng.integers(0, 256, size=(gh, gw)).astype(np.float64)
    img = np.kron(cells, np.ones((block, block)))[:h, :w]
    img = img + rng.normal(0.0, noise, size=(h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _snapshot(prefix, feats, out):
    out[prefix + "_keypoints"] = feats.keypoints.copy()
    out[prefix + "_landmarks"] = feats.landmarks.copy()
    out[prefix + "_state"] = np.asarray(feats.state, dtype=np.float64).copy()
    out[prefix + "_tracks"] = feats.tracks.copy()
    out[prefix + "_poses"] = feats.poses.copy()
    out[prefix + "_candidate_mask"] = feats.candidate_mask.copy()


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def bookkeeping_scenario(ns):
    """3-frame scripted run through Features/Frame/Matches/State/Camera and
    LandmarksTriangulator.triangulate_candidates (use_opencv=False).

    ``ns`` supplies the classes: Features, Frame, Matches, State, Camera,
    LandmarksTriangulator.
    """
    out = {}
    rng = np.random.default_rng(77)
    K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
    n = 40
    X = rng.uniform(-2, 2, size=(n, 3, 1))
    X[:, 2] = X[:, 2] * 2 + 12

    # world -> camera transforms for three frames (camera moves sideways + yaw)
    def c_T_w(k):
        T = np.eye(4)
        T[:3, :3] = _rot_y(0.02 * k)
        T[:3, 3] = [-0.6 * k, 0.05 * k, 0.1 * k]
        return T

    def project(k):
        T = c_T_w(k)
        Xc = T[:3, :3][None] @ X + T[:3, 3:].reshape(1, 3, 1)
        u = K[None] @ Xc
        return u[:, :2] / u[:, 2:]

    cam = ns.Camera(intrinsic_matrix=K)
    kp0, kp1, kp2 = project(0), project(1), project(2)

    f0 = ns.Frame(None, features=ns.Features(keypoints=kp0.copy()), sensor=cam, intrinsics=K)
    state = ns.State(f0, bearing_threshold=0.05)

    # ---- bootstrap step (main.py:204-230), GT relative pose ----
    # frame 1 detects a shuffled subset + 5 spurious points
    perm1 = rng.permutation(n)[:32]
    extra1 = rng.uniform(0, 400, size=(5, 2, 1))
    f1 = ns.Frame(None, features=ns.Features(keypoints=np.concatenate([kp1[perm1], extra1])),
                  sensor=cam, intrinsics=K)
    pairs01 = np.stack([perm1, np.arange(32)], axis=-1)
    pairs01 = pairs01[rng.permutation(32)[:28]]          # 28 matches, arbitrary order
    m01 = ns.Matches(f0, f1, pairs01)
    _snapshot("m01_f1", m01.frame1.features, out)
    _snapshot("m01_f2", m01.frame2.features, out)
    state.update_from_matches(m01)

    M = c_T_w(1)[:3]                                     # camera0 -> camera1
    n_m = int(np.sum(m01.frame2.features.match_inliers))
    inl = np.ones(n_m, dtype=bool)
    inl[[3, 11]] = False                                 # two bootstrap outliers
    # landmarks in camera-0 (= world) frame for the matched points, frame-2 order
    idx_world = pairs01[:, 0]
    lm = X[idx_world]
    outliers = np.zeros(m01.frame2.features.length, dtype=bool)
    outliers[m01.frame2.features.match_inliers] = ~inl
    state.update_with_local_pose(M)
    mask = np.zeros_like(m01.frame2.features.matched_candidate_inliers).astype(bool)
    mask[m01.frame2.features.matched_candidate_inliers] = inl
    state.update_with_local_landmarks(lm[inl], mask)
    state.reset_outliers(outliers)
    out["boot_pose"] = state.get_pose().copy()
    _snapshot("boot", state.curr_frame.features, out)

    # ---- steady-state steps (main.py:248-286), GT world poses ----
    def owner_of(kp_cur, kp_gt):
        d = np.linalg.norm(kp_cur[:, None, :, 0] - kp_gt[None, :, :, 0], axis=-1)
        return d.argmin(axis=1), d.min(axis=1) < 1e-9

    kp_gt = {1: kp1, 2: kp2, 3: project(3)}
    tri = ns.LandmarksTriangulator(camera1=cam, camera2=cam, use_ransac=False, use_opencv=False)
    for k in (2, 3):
        tag = "s%d" % k
        nf = state.curr_frame.features.length
        keep = np.sort(rng.permutation(nf)[: nf - 4])
        owner, is_real = owner_of(state.curr_frame.features.keypoints, kp_gt[k - 1])
        sel = [i for i in keep if is_real[i]]
        kp_new = np.concatenate([kp_gt[k][owner[sel]], rng.uniform(0, 400, size=(6, 2, 1))])
        fk = ns.Frame(None, features=ns.Features(keypoints=kp_new), sensor=cam, intrinsics=K)
        pairs = np.stack([np.array(sel), np.arange(len(sel))], axis=-1)
        pairs = pairs[rng.permutation(len(sel))]
        m = ns.Matches(state.curr_frame, fk, pairs)
        _snapshot(tag + "_m_f1", m.frame1.features, out)
        _snapshot(tag + "_m_f2", m.frame2.features, out)

        n_tri = int(np.sum(m.frame2.features.triangulate_inliers))
        p3p_inl = np.ones(n_tri, dtype=bool)
        p3p_inl[[1]] = False
        outliers = np.zeros(m.frame2.features.length, dtype=bool)
        outliers[m.frame2.features.triangulate_inliers] = ~p3p_inl
        state.update_from_matches(m)
        state.update_with_world_pose(c_T_w(k)[:3])
        state.reset_outliers(outliers)
        state.compute_candidates()
        out[tag + "_pose"] = state.get_pose().copy()
        _snapshot(tag + "_pre", state.curr_frame.features, out)
        kc = state.curr_frame.features
        n_mc = kc.matched_candidate_inliers_poses.shape[0]
        out[tag + "_bearing"] = state._calculate_bearing_angle(
            cam,
            kc.matched_candidate_inliers_poses,
            np.stack([state.curr_pose] * n_mc, axis=0),
            kc.matched_candidate_inliers_tracks,
            kc.matched_candidate_inliers_keypoints,
        ) if n_mc > 0 else np.zeros(0)
        if np.sum(kc.candidate_mask) > 0:
            Xw = tri.triangulate_candidates(kc, current_pose=state.get_pose())
            out[tag + "_cand_landmarks"] = Xw.copy()
            state.update_with_world_landmarks(Xw, m.frame2.features.candidate_mask)
        _snapshot(tag + "_post", state.curr_frame.features, out)
    return out
