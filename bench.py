#!/usr/bin/env python3
"""Benchmark: VO frames/s on the BASELINE.json config-2 workload, through the device-resident frame loop.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU under torch.distributed.run; started that way by the driver, or by this script
   itself when it is called directly with --gpus N)

The stream is SURVEY.md 8d's: 100 frames of a strictly forward drive (0.8 m per frame, src/main.py:248), all resident
in HBM, walked 2 -> 99 pass after pass.  At the end of a pass the look-ahead is drained and the state the bootstrap
handed over for frame 2 is put back on the device (vo_pipeline_rewind: one device-to-device copy, that frame's pyramid
and detection queued behind it, no host synchronisation; the estimator's RANSAC fields and generator go on) -- the seam
is inside the timed region.  Tracks leave the field of view as the camera advances, so the re-detect of
klt.py:207-230 fires at the rate the stream produces and the detector executes when it is needed.

One "step" = one frame of the reference's steady-state loop (src/main.py:248-286, KLT tracker mode with the
Harris detector) on a 1376x1241 frame that is already resident in HBM, everything on the GPU:
  pyramid(next) | Harris response + exact greedy NMS (2000 keypoints) on next, for a sequence whose track count is
  within 1 % (+ 2.5 times its last loss) of the re-detect limit (the reference runs its detector only below the limit, klt.py:207-230;
  VO_BENCH_DETECT_MARGIN=-1: on every frame) | re-detect append when fewer than 80 % of the tracks survive -> KLT
  (3 levels, 15x15) of every feature -> Matches regroup -> P3P-RANSAC (1000
  hypotheses solved + scored, reference-exact sampler, sequential accept rule replayed on the device) -> pose
  refinement over the inliers -> State bookkeeping (reset_outliers, bearing-angle candidates) -> DLT of the
  candidates with one start pose per track -> landmark insertion + cheirality check.
The Features / State / RANSAC bookkeeping never leaves HBM; the landmarks every pose is estimated from are the
ones the loop itself triangulated (after a real two-view bootstrap on the host, untimed).  Each rank runs its own
synthetic sequence (weak scaling, frame streams shard at sequence granularity); with N > 1 the ranks' {pose,
landmarks} records are all-gathered over RCCL on a side stream.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

# The GPU boxes cap a job at 16 CPUs' worth of time per 100 ms (cgroup cpu.max) on a 256-core host: thread pools
# sized for 256 cores (OpenBLAS / OpenMP under NumPy and torch) overrun that quota in bursts and the whole job is
# frozen for the rest of the period -- seen as single 40-80 ms gaps in the timed region (cpu.stat nr_throttled).
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "4")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "visual-odometry-project_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

H, W, N_KP, HYP, WIN, MAX_LEVEL = 1241, 1376, 2000, 1000, 15, 2
CONFIG = os.environ.get("VO_BENCH_CONFIG", "cfg2")
if CONFIG == "cfg5":     # BASELINE.json configs[4]: the stress shape (secondary line under profiles/, the default stays cfg-2)
    H, W, N_KP, HYP, MAX_LEVEL = 2160, 3840, 8000, 4000, 3
# Hypotheses solved + scored per launch: the RANSAC budget (max_iterations = HYP, ransac.py:90: `while n < n_iterations`
# counts iterations, and an iteration whose P3P has no solution is not counted, ransac.py:98-101) needs 3-4 % more samples
# than iterations on this stream -- a launch holds the budget plus 1/8
HYP_LAUNCH = int(os.environ.get("VO_BENCH_HYP_LAUNCH", str(HYP + HYP // 8 + 24)))
N_FRAMES = 30 if CONFIG == "cfg5" else 100    # cfg-2 100 frames, cfg-5 30 frames
PASS_START = 2           # the bootstrap uses frames 0 and 2 (main.py:204-209); a pass walks PASS_START -> N_FRAMES - 1
S_LEG = 16               # sequences per GPU of the in-line throughput leg
# frames of the first pass that also go through the CPU oracle of the loop (parity + CPU baseline): the whole pass --
# its re-detect frames included -- is 97 frames, 12-15 s of one core
CPU_BASELINE_FRAMES = int(os.environ.get("VO_BENCH_CPU_FRAMES", "97"))
RENDER_WORKERS = int(os.environ.get("VO_BENCH_RENDER_WORKERS", str(max(1, min(12, (os.cpu_count() or 2) - 2)))))
REFINE_ITERS = int(os.environ.get("VO_BENCH_REFINE", "20"))   # Gauss-Newton steps allowed to the pose refinement (0: off)
EXCHANGE_EVERY = int(os.environ.get("VO_BENCH_EXCHANGE_EVERY", "16"))   # frames per all-gather of {pose, landmarks} records
REDETECT_POSE = os.environ.get("VO_BENCH_REDETECT_POSE", "current")   # see vo_pipeline_config.redetect_start_pose
PREPARE_AHEAD = os.environ.get("VO_BENCH_PREPARE", "1") != "0"      # vo_pipeline_prepare: the next frame's pyramid one step ahead
DETECT_MARGIN = float(os.environ.get("VO_BENCH_DETECT_MARGIN", "0.01"))   # see vo_pipeline_config.detect_margin (< 0: every frame)
PROF_EVERY = 4           # HIP-event pairs around every 4th launch of the dominant kernel in the timed region
HBM_PEAK_GBS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
PROFILE_TAG = os.environ.get("VO_BENCH_PROFILE_TAG", "r03")


def algorithmic_bytes(kernel_name, n_in, n_tracked, n_tri):
    """Compulsory HBM traffic per launch (DESIGN.md 'Kernels'): every input read once,
    every required output written once, at the feature counts the run actually had."""
    px = H * W
    levels = MAX_LEVEL + 1
    feat = 8 + 16 + 2 + 24 + 16 + 96                          # one Features row: kp f32, kp f64, state, cand, landmark, track, pose
    table = {
        "harris_response": px * 1 + px * 8,                   # image read once, fp64 map written once
        "nms_candidates": px * 8,                             # fp64 map read once
        "nms_compact": N_KP * 12,
        "nms_round": N_KP * 4 * 2,
        "nms_rank": N_KP * 12,
        "nms_select": N_KP * 16,
        "pyr_down": 2 * px + px // 4 + px // 16,              # frame read, bordered copy of level 0, levels 1 and 2 written
        "klt_track": n_in * levels * ((WIN + 3) ** 2 + (WIN + 1) ** 2) + n_in * (8 + 8 + 1 + 4),
        "state_regroup": n_in * (8 + 1 + 4 + 1) + n_tracked * 2 * feat,
        "p3p_solve": HYP_LAUNCH * (28 + 4 * 40 + 96 + 1) + n_tri * 40 + HYP_LAUNCH * (4 + ((n_tri + 63) // 64) * 8),   # solve + score in one launch
        "p3p_score": n_tri * 40 + HYP * (96 + 1 + 4 + ((n_tri + 63) // 64) * 8),
        "ransac_replay": HYP_LAUNCH * 5 + 96 + ((n_tri + 63) // 64) * 16,
        "refine_pose": n_tri * 40 + ((n_tri + 63) // 64) * 8 + 96 + 120,
        "state_candidates": n_tracked * (16 + 1 + 16 + 96 + 1) + ((n_tri + 63) // 64) * 8,
        "state_landmarks": n_tracked * (1 + 1 + 24 + 16 + 96 + 24),
    }
    # the frame loop's pose kernel (kernel id "refine_pose"): replay + refinement in one launch of one workgroup; its walk
    # (candidates) and the landmark stage in one launch of cap / 256 workgroups (kernel id "state_landmarks")
    table["refine_pose"] = table["ransac_replay"] + table["refine_pose"]
    table["state_landmarks"] = table["state_candidates"] + table["state_landmarks"]
    return table.get(kernel_name)


KERNEL_LABEL = {"refine_pose": "frame_pose (RANSAC replay + pose refinement, one workgroup)",
                "state_landmarks": "state_walk_landmarks (State bookkeeping of every feature + candidate DLT + cheirality + record)",
                "p3p_solve": "p3p_hyp (P3P solve + inlier counts of all hypotheses)"}
ROCPROF_NAMES = {"klt_track": "klt_track16_kernel<15, 16>", "nms_round": "nms_round_kernel<5, true>",
                 "nms_candidates": "nms_candidates_kernel<5>", "harris_response": "harris_response_kernel<9>",
                 "p3p_solve": "p3p_hyp_kernel<4>", "p3p_score": "p3p_score_kernel", "nms_compact": "nms_compact_kernel",
                 "nms_rank": "nms_rank_kernel", "nms_select": "nms_finalize_kernel", "refine_pose": "frame_pose_kernel",
                 "pyr_down": "pyramid3_tiled_kernel<32, 16>", "state_regroup": "state_regroup_klt_kernel<false>",
                 "ransac_replay": "ransac_replay_kernel", "state_candidates": "state_candidates_kernel",
                 "state_landmarks": "state_walk_landmarks_kernel"}


def pmc_traffic(kernel_name):
    """HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (profiles/<tag>_pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate runs of this same command, FETCH_SIZE doubled for gfx950 as
    MI355X_MICROARCH.md prescribes); None if the file or the kernel is missing.  bench.py itself cannot run under
    the counters."""
    try:
        with open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_traffic.json")) as f:
            table = json.load(f)
        return int(table[ROCPROF_NAMES[kernel_name]]["hbm_bytes_corrected"])
    except (OSError, KeyError, ValueError):
        return None


class ResidentSequence:
    """The frames of a synthetic.Stream as the iterator vo.driver.bootstrap expects (Sequence surface)."""

    def __init__(self, stream):
        from vo.sensors import Camera
        self.stream, self.idx = stream, 0
        self.camera = Camera(intrinsic_matrix=stream.K)

    def get_camera(self):
        return self.camera

    def __iter__(self):
        return self

    def __next__(self):
        from vo.primitives import Frame
        if self.idx >= self.stream.n:
            raise StopIteration
        f = Frame(self.stream.image(self.idx), sensor=self.camera, intrinsics=self.stream.K)
        f.frame_id = self.idx
        self.idx += 1
        return f


class Walker:
    """Drives one pipeline over the resident stream: frames PASS_START -> n_frames - 1, then the seam (drain, rewind
    to the checkpoint taken at PASS_START), pass after pass; at most two steps in flight (one frame of look-ahead)."""

    def __init__(self, pipe, n_frames):
        self.pipe, self.n_frames, self.cur, self.passes = pipe, n_frames, PASS_START, 0

    def run(self, n, lookahead=True, on_step=None):
        """n steps; on_step(next_frame_index, [StepResult per sequence]) after each collect."""
        pipe, flight, submitted, done = self.pipe, [], 0, 0
        depth = 2 if lookahead else 1
        while done < n:
            while submitted < n and len(flight) < depth:
                if self.cur + 1 >= self.n_frames:              # the seam
                    if flight:
                        break                                  # (drain first: rewind wants nothing in flight)
                    pipe.rewind()
                    self.cur = PASS_START
                    self.passes += 1
                pipe.submit(self.cur, self.cur + 1)
                if PREPARE_AHEAD and self.cur + 2 < self.n_frames:
                    pipe.prepare(self.cur + 2)                 # the coming step's pyramid, behind this step's tracker
                flight.append(self.cur + 1)
                self.cur += 1
                submitted += 1
            rs = pipe.collect_all()
            b = flight.pop(0)
            done += 1
            if on_step is not None:
                on_step(b, rs)


def bootstrap_state(stream):
    """main.py:204-230 on frames 0 and 2 of the stream, on the host (untimed): Shi-Tomasi corners (as many as the
    detector keeps per frame), KLT, 8-point RANSAC, essential matrix, cheirality, DLT.  Two settings differ from
    the per-frame loop because frames 0 and 2 are two steps apart at this resolution (measured, tools/dev/boot_dbg.py:
    with the loop's 15x15 / 3-level tracker and the reference's 0.25 px epipolar threshold the 8-point RANSAC settles
    on a wrong model, 114 landmarks, 3 m off): the bootstrap tracks with 21x21 / 4 levels and accepts 1 px."""
    from vo import driver
    from vo.features.klt import KLTTracker
    saved = (dict(KLTTracker._feature_params), dict(KLTTracker._lk_params))

    def setup():
        KLTTracker._feature_params = dict(saved[0], maxCorners=N_KP)
        KLTTracker._lk_params = dict(saved[1], winSize=(21, 21), maxLevel=3)

    try:
        state, tracker, _, _ = driver.bootstrap(ResidentSequence(stream), "klt", tracker_setup=setup, ransac_threshold=1.0)
    finally:
        KLTTracker._feature_params, KLTTracker._lk_params = saved
    return state


def oracle_leg(stream, state, gpu_results, gpu_state, state_frames):
    """The CPU oracle of the same loop (tests/pipeline_oracle.py: pinned bookkeeping classes + CPU oracles) on the
    first len(gpu_results) frames of the pass: (i) parity of the GPU's records against it, and of the GPU's Features
    arrays after `state_frames` frames, (ii) its time = the CPU baseline ("port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import harris_np
    from pipeline_oracle import OracleLoop
    frames = len(gpu_results)
    orc = OracleLoop(stream, N_KP, WIN, MAX_LEVEL, refine_iters=REFINE_ITERS, redetect_start_pose=REDETECT_POSE)
    orc.set_state(PASS_START, state.curr_frame.features, state.curr_pose, state.prev_pose)
    order = list(range(PASS_START, PASS_START + frames + 1))
    dR = dt = 0.0
    exact = True
    t0 = time.perf_counter()
    refs = []
    f_state = None
    for b in order[1:]:
        refs.append(orc.step(b))
        if len(refs) == state_frames:
            import copy
            f_state = copy.deepcopy(refs[-1]["features"])     # (the next step's Matches regroups this object in place)
        # the reference's tracker runs its detector only when it re-detects (klt.py:207-230); the oracle loop does that
        # inside step().  (Rounds 1-2 added one NumPy Harris + NMS per frame here, matching a GPU step that detected on
        # every frame; the GPU loop no longer does.)
    cpu_s = time.perf_counter() - t0
    n_redetect = int(sum(1 for ref in refs if ref["n_before"] < 0.8 * N_KP))      # klt.py:208-212
    for ref, r in zip(refs, gpu_results):
        Rr, tr = np.array(r.R_refined).reshape(3, 3), np.array(r.t_refined)
        dR = max(dR, float(np.abs(Rr - ref["R_ref"]).max()))
        dt = max(dt, float(np.abs(tr - ref["t_ref"]).max() / max(1.0, np.linalg.norm(ref["t_ref"]))))
        exact &= (r.n_tracked, r.n_triangulated, r.n_inliers, r.draws_consumed, r.ransac_iterations, r.n_candidates,
                  r.n_landmarks) == (ref["n_tracked"], ref["n_tri"], ref["n_inliers"], ref["draws"], ref["iters"],
                                     ref["n_cand"], ref["n_landmarks"])
    f = f_state
    exact &= bool(np.array_equal(gpu_state["keypoints"], f.keypoints.astype(np.float64)) and
                  np.array_equal(gpu_state["state"], f.state) and
                  np.array_equal(gpu_state["candidate_mask"], f.candidate_mask) and
                  np.array_equal(gpu_state["tracks"], f.tracks, equal_nan=True))
    with np.errstate(invalid="ignore"):
        d = np.abs(gpu_state["landmarks"] - f.landmarks)[:, :, 0].max(axis=1)
        lm = float(np.nanmax(d / np.maximum(1.0, np.linalg.norm(f.landmarks[:, :, 0], axis=1)))) if f.length else 0.0
    parity = {"frames": frames, "redetect_frames": n_redetect, "max_abs_dR": dR, "max_rel_dt": dt,
              "max_rel_dlandmark_after_%d_frames" % state_frames: lm,
              "counts_masks_keypoints_states_tracks_exact": bool(exact),
              "note": "refined pose of every frame / Features arrays after %d frames vs the CPU oracle of the same loop on the same "
                      "frames (tolerance of the metric: 1e-4 rel.); integer and index results must be identical" % state_frames}
    base = {"value": frames / cpu_s, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "the first %d frames of the same stream through the CPU oracle of the loop (tests/pipeline_oracle.py: "
                      "C KLT / P3P, NumPy RANSAC / refinement / DLT / bookkeeping; NumPy Harris + the oracle's fast exact NMS walk "
                      "on the %d frame(s) that re-detect), single thread, %.1f s" % (frames, n_redetect, cpu_s)}
    return parity, base


def api_leg(ctx):
    """Frames/s through the Python drop-in API at the same frame size: (a) the reference-shaped classes, one call per
    stage with host bookkeeping (vo.driver.run); (b) the same loop on the device pipeline with one image upload per
    frame (vo.driver.run_on_device)."""
    from vo import driver
    from vo.primitives import Sequence
    out = {}
    try:
        seq = Sequence("synthetic", n_frames=9, height=H, width=W, channels=1)
        r = driver.run(seq, "klt")
        out["drop_in_classes_frames_per_s"] = round(float(1.0 / np.median(r["frame_seconds"])), 1)
        seq = Sequence("synthetic", n_frames=9, height=H, width=W, channels=1)
        t0 = time.perf_counter()
        r = driver.run_on_device(seq, n_keypoints=N_KP, klt_win=WIN, klt_max_level=MAX_LEVEL, hyp=4 * HYP, context=ctx,
                                 bootstrap_win=21, bootstrap_max_level=3, bootstrap_threshold=1.0)   # as bootstrap_state above
        out["device_pipeline_steps_finished_by_host_path"] = int(sum(x.recovered for x in r["results"]))
        out["device_pipeline_with_upload_frames_per_s"] = round(float(1.0 / np.median(r["frame_seconds"])), 1)
        out["note"] = ("640x480-independent: both at 1376x1241; (a) vo.driver.run = reference call order through Tracker / "
                       "P3PPoseEstimator / LandmarksTriangulator / State with host arrays between stages, Shi-Tomasi 500 "
                       "corners as the reference configures KLT; (b) vo.driver.run_on_device = host bootstrap, then one "
                       "1.7 MB image upload + one submit per frame, 2000 keypoints, 4000 hypotheses per launch, main.py's P3P "
                       "settings (1.25 px, confidence 0.9999, up to 10000 iterations: a frame that needs more than one launch "
                       "of hypotheses gets further ones, on the device)")
    except Exception as e:                                   # the headline must not depend on this leg
        out["error"] = repr(e)
    return out


def cfg3_main(args, tracker="sift"):
    """VO_BENCH_CONFIG=cfg3: BASELINE.json configs[2] -- the same stream with SIFT detect + describe (cap 2000) and
    brute-force L2 2-NN + ratio + uniqueness matching in place of KLT (src/vo/features/tracker.py:60-61, sift.py:23-56),
    as a device-resident loop (vo_pipeline_config.tracker_mode = 1): images in, pose records out; descriptors, pair
    lists, Features and State never leave HBM.  A secondary line: the default run stays cfg-2."""
    from vo import synthetic
    stream = synthetic.Stream(N_FRAMES, H, W).prefetch(workers=RENDER_WORKERS)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    from vo import _native, driver
    from vo.features.harris import HarrisCornerDetector
    from vo.features.sift import SIFTDetector
    ctx = _native.Context(0)
    _native.set_default_context(ctx)
    CAP = 2000
    saved = (SIFTDetector._max_keypoints, HarrisCornerDetector._num_keypoints_override)
    SIFTDetector._max_keypoints = CAP
    HarrisCornerDetector._num_keypoints_override = CAP
    try:
        state, _, _, _ = driver.bootstrap(ResidentSequence(stream), tracker, ransac_threshold=1.0)
    finally:
        SIFTDetector._max_keypoints, HarrisCornerDetector._num_keypoints_override = saved
    feats = state.curr_frame.features
    pipe = _native.Pipeline(ctx, H, W, N_FRAMES, stream.K, n_keypoints=CAP, hyp=HYP_LAUNCH, p3p_threshold=1.0, outlier_ratio=0.9,
                            confidence=0.99, max_iterations=HYP, refine_iters=REFINE_ITERS, tracker=tracker, sift_cap=CAP)
    for i in range(N_FRAMES):
        pipe.set_frame(i, stream.image(i))
    pipe.set_state(PASS_START, feats, state.curr_pose, state.prev_pose, num_features=CAP)
    pipe.checkpoint()
    walker = Walker(pipe, N_FRAMES)
    first_pass = []
    ORACLE_FRAMES = 3
    walker.run(ORACLE_FRAMES, lookahead=False, on_step=lambda b, rs: first_pass.append((b, rs[0])))
    first_state = pipe.get_state()
    walker.run(N_FRAMES - 1 - PASS_START - ORACLE_FRAMES, lookahead=False, on_step=lambda b, rs: first_pass.append((b, rs[0])))
    walker.run(args.warmup)
    ctx.prof_enable(-1)
    pipe.prof_reset()
    walker.run(8)
    per = {}
    for kid in range(_native.K_COUNT):
        ms, n = pipe.prof_read(kid)
        if n:
            per[ctx.kernel_name(kid)] = ms / n * 1e3
    ctx.prof_disable()
    recs = []
    ctx.sync()
    t0 = time.perf_counter()
    walker.run(args.steps, on_step=lambda b, rs: recs.extend(rs))
    ctx.sync()
    dt = time.perf_counter() - t0
    st = loop_stats(recs)
    # ground truth over the first pass; oracle of the same loop on its first frames (also the CPU baseline)
    scale = 2 * synthetic.STEP_Z / max(np.linalg.norm(state.curr_pose[:3, 3]), 1e-12)
    gt_err = []
    for b, r in first_pass:
        Twc = r.pose_world_cam()
        gt = np.linalg.inv(stream.T_world_cam(0)) @ stream.T_world_cam(b)
        gt_err.append((float(np.linalg.norm(Twc[:3, :3] - gt[:3, :3])), float(np.linalg.norm(scale * Twc[:3, 3] - gt[:3, 3]))))
    out = {"metric": "VO frames/sec at 1376x1241, 2k keypoints; pose err vs reference", "value": round(args.steps / dt, 2),
           "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32 (SIFT) / u8 -> i32 (matching) / f64 (pose)", "data": "synthetic",
           "config": {"workload": ("cfg-3: %dx%d synthetic stream, %d frames strictly forward, per frame SIFT detect + describe (%d "
                                   "strongest), brute-force L2 2-NN + 0.8 ratio + uniqueness against the current Features' "
                                   "descriptors (v_mfma_i32_32x32x32_i8), Matches regroup from the pair list, P3P-RANSAC %d "
                                   "iterations + device replay, pose refinement, State bookkeeping, candidate DLT; device-resident "
                                   "(vo_pipeline tracker_mode = sift), frames resident in HBM" if tracker == "sift" else
                                   "Tracker(mode='harris') on the cfg-2 stream (%dx%d, %d frames strictly forward): per frame Harris "
                                   "response + exact greedy NMS (%d keypoints), raw 19x19 patches as bytes, brute-force L2 2-NN + 0.85 "
                                   "ratio + uniqueness against the current Features' patches (v_mfma_i32_32x32x32_i8, 361 values), "
                                   "Matches regroup from the pair list, P3P-RANSAC %d iterations + device replay, refinement, State "
                                   "bookkeeping, candidate DLT; device-resident (vo_pipeline tracker_mode = harris)") % (W, H, N_FRAMES, CAP, HYP),
                      "keypoints": CAP, "frames_resident": N_FRAMES, "frame_lookahead": 1,
                      "passes_started_in_timed_region": walker.passes},
           "loop": dict(st, bootstrap_landmarks=int((feats.state == 2).sum())),
           "per_kernel_us": {k: round(v, 1) for k, v in sorted(per.items())},
           "pose_err_vs_ground_truth": {"rot_fro_median": float(np.median([e[0] for e in gt_err])),
                                        "trans_m_median": float(np.median([e[1] for e in gt_err])),
                                        "trans_m_last": round(gt_err[-1][1], 3),
                                        "trans_m_every_8th_frame": [round(e[1], 3) for e in gt_err[::8]],
                                        "path_m": round((N_FRAMES - 1 - PASS_START) * synthetic.STEP_Z, 1), "frames": len(gt_err)}}
    ss_us = per.get("sift_scale_space")
    if ss_us:
        ss_bytes = (4 * H * W * 4) * (4.0 / 3.0) * 22
        out["roofline"] = {"bound": "hbm", "kernel": "sift_scale_space (upsample, fused row+column blurs, decimations; the octaves' "
                           "last two layers and extrema searches on further streams, inside the same bracket)",
                           "avg_launch_us": round(ss_us, 1), "algorithmic_bytes_per_launch": int(ss_bytes),
                           "achieved": round(ss_bytes / (ss_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ss_bytes / (ss_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None}
    m_us = per.get("match_knn2")
    if m_us:
        flops = 2.0 * CAP * CAP * 128
        out["matcher"] = {"bound": "mfma", "kernel": "match_knn2 (v_mfma_i32_32x32x32_i8 on bytes offset by 128, exact integer distances)",
                          "avg_launch_us": round(m_us, 1), "ops": flops, "achieved": round(flops / (m_us * 1e-6) / 1e12, 2),
                          "peak": 5000.0, "unit": "TOP/s", "frac": round(flops / (m_us * 1e-6) / 1e12 / 5000.0, 5),
                          "note": "2000 x 2000 x 128: 1 GOP -- latency-bound, not MFMA-bound"}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from pipeline_oracle import OracleLoop
        orc = OracleLoop(stream, CAP, WIN, MAX_LEVEL, refine_iters=REFINE_ITERS, tracker=tracker)
        orc.set_state(PASS_START, feats, state.curr_pose, state.prev_pose)
        t1 = time.perf_counter()
        refs = [orc.step(PASS_START + 1 + k) for k in range(ORACLE_FRAMES)]
        cpu_s = time.perf_counter() - t1
        dR = dtr = 0.0
        exact = True
        for ref, (_, r) in zip(refs, first_pass):
            dR = max(dR, float(np.abs(np.array(r.R_refined).reshape(3, 3) - ref["R_ref"]).max()))
            dtr = max(dtr, float(np.abs(np.array(r.t_refined) - ref["t_ref"]).max() / max(1.0, np.linalg.norm(ref["t_ref"]))))
            exact &= (r.n_tracked, r.n_triangulated, r.n_inliers, r.draws_consumed, r.ransac_iterations, r.n_candidates,
                      r.n_landmarks) == (ref["n_tracked"], ref["n_tri"], ref["n_inliers"], ref["draws"], ref["iters"],
                                         ref["n_cand"], ref["n_landmarks"])
        f = refs[-1]["features"]
        exact &= bool(np.array_equal(first_state["keypoints"], f.keypoints.astype(np.float64)) and
                      np.array_equal(first_state["state"], f.state) and
                      np.array_equal(first_state["candidate_mask"], f.candidate_mask))
        out["pose_vs_oracle"] = {"frames": ORACLE_FRAMES, "max_abs_dR": dR, "max_rel_dt": dtr,
                                 "counts_keypoints_states_masks_exact": bool(exact),
                                 "note": ("vs the CPU oracle of the same loop (oracle/csrc/sift.c, match.c, p3p.c + the pinned "
                                          "bookkeeping classes); parity unpinned vs OpenCV's SIFT / BFMatcher" if tracker == "sift" else
                                          "vs the CPU oracle of the same loop (oracle/harris_np.py pinned to the reference, "
                                          "oracle/csrc/match.c, p3p.c + the pinned bookkeeping classes); parity unpinned vs "
                                          "OpenCV's BFMatcher")}
        out["cpu_baseline"] = {"value": ORACLE_FRAMES / cpu_s, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d frames through the CPU oracle of the loop, single thread, %.1f s" % (ORACLE_FRAMES, cpu_s)}
    _Stdout.emit(out)
    pipe.close()
    ctx.close()


def spawn_ranks(args):
    """Called with --gpus N > 1 outside torch.distributed.run: start the N ranks as children (before anything
    touches the GPU in this process) and relay rank 0's line."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" % (args.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29531"),
           os.path.abspath(__file__)] + sys.argv[1:]
    # (this process's descriptor 1 points at stderr, see _Stdout: the ranks get the real stdout)
    raise SystemExit(subprocess.run(cmd, stdout=_Stdout.file.fileno() if _Stdout.file else None).returncode)


def make_pipeline(ctx, streams, states, S, detect_margin, hyp=None):
    """A pipeline over the resident streams of S sequences, every frame uploaded, the bootstraps' states handed over
    for frame PASS_START and checkpointed there (the seam of every later pass)."""
    from vo import _native
    pipe = _native.Pipeline(ctx, H, W, N_FRAMES, streams[0].K, n_keypoints=N_KP, klt_win=WIN, klt_max_level=MAX_LEVEL,
                            hyp=hyp or HYP_LAUNCH, p3p_threshold=1.0, outlier_ratio=0.9, confidence=0.99, max_iterations=HYP,
                            refine_iters=REFINE_ITERS, redetect_start_pose=REDETECT_POSE, sequences=S,
                            detect_margin=detect_margin, debug_never_detect=int(os.environ.get("VO_BENCH_NEVER_DETECT", "0")))
    for q in range(S):
        for i in range(N_FRAMES):
            pipe.set_frame(i, streams[q].image(i), seq=q)
        pipe.set_state(PASS_START, states[q].curr_frame.features, states[q].curr_pose, states[q].prev_pose,
                       num_features=N_KP, seq=q)
    pipe.checkpoint()
    return pipe


def upload_leg(ctx, stream, state):
    """The headline's loop with the frames coming from HOST memory, one 1.7 MB upload per frame (PCIe-inclusive: never
    `value`): a pipeline with four frame slots walks the stream once, frame PASS_START -> N_FRAMES - 1, in three ways --
    (a) frames in pinned memory (vo_host_alloc: where a frame grabber or a decoder would put them), uploaded one step
    ahead of their use by vo_pipeline_set_frame_pinned (DMA on its own stream, beside the kernels); (b) the same, each
    frame uploaded only when its own step is submitted; (c) frames in ordinary NumPy arrays through vo_pipeline_set_frame
    (staging copy on the host, DMA in front of the frame's pyramid on the tracker's stream)."""
    from vo import _native
    n = stream.n
    pinned = [ctx.pinned_empty((H, W)) for _ in range(n)]
    plain = [np.ascontiguousarray(stream.image(i)) for i in range(n)]
    for i in range(n):
        pinned[i][...] = plain[i]
    out = {}
    for name, src, ahead in (("pinned_uploaded_one_step_ahead", pinned, 1), ("pinned_uploaded_with_its_step", pinned, 0),
                             ("pageable_numpy_array", plain, 0)):
        pipe = _native.Pipeline(ctx, H, W, 4, stream.K, n_keypoints=N_KP, klt_win=WIN, klt_max_level=MAX_LEVEL, hyp=HYP_LAUNCH,
                                p3p_threshold=1.0, outlier_ratio=0.9, confidence=0.99, max_iterations=HYP,
                                refine_iters=REFINE_ITERS, redetect_start_pose=REDETECT_POSE, detect_margin=DETECT_MARGIN)
        rate = None
        for timed in (False, True):                           # a warm pass, then the timed one from the same state
            pipe.set_frame(PASS_START % 4, src[PASS_START], pinned=src is pinned)
            pipe.set_state(PASS_START % 4, state.curr_frame.features, state.curr_pose, state.prev_pose, num_features=N_KP)
            if ahead:
                pipe.set_frame((PASS_START + 1) % 4, src[PASS_START + 1], pinned=src is pinned)
            ctx.sync()
            recs, pending = [], 0
            is_pinned = src is pinned
            t_set = t_sub = t_col = 0.0
            t0 = time.perf_counter()
            for k in range(PASS_START, n - 1):                # step k: frame k -> k + 1
                ta = time.perf_counter()
                if pending == 2:
                    recs.append(pipe.collect())
                    pending -= 1
                tb = time.perf_counter()
                if ahead:
                    if k + 2 < n:
                        pipe.set_frame((k + 2) % 4, src[k + 2], pinned=is_pinned)
                else:
                    pipe.set_frame((k + 1) % 4, src[k + 1], pinned=is_pinned)
                tc = time.perf_counter()
                pipe.submit(k % 4, (k + 1) % 4)
                if PREPARE_AHEAD and ahead and k + 2 < n:
                    pipe.prepare((k + 2) % 4)
                pending += 1
                td = time.perf_counter()
                t_col += tb - ta
                t_set += tc - tb
                t_sub += td - tc
            while pending:
                recs.append(pipe.collect())
                pending -= 1
            dt = time.perf_counter() - t0
            if timed:
                rate = (n - 1 - PASS_START) / dt
                out[name + "_host_path_steps"] = int(sum(r.recovered for r in recs))
                ts = np.array([[r.ts[j] for j in range(6)] for r in recs], dtype=np.float64) * 1e-2
                out[name + "_chain_us"] = {"step_period": round(float(np.median(np.diff(ts[:, 1]))), 1),
                                           "tracker_start_to_regroup_start": round(float(np.median(ts[:, 1] - ts[:, 0])), 1),
                                           "regroup_to_next_tracker_start": round(float(np.median(ts[1:, 0] - ts[:-1, 1])), 1),
                                           "regroup_to_record": round(float(np.median(ts[:, 5] - ts[:, 1])), 1),
                                           "record_to_next_regroup": round(float(np.median(ts[1:, 1] - ts[:-1, 5])), 1),
                                           "detector_executed": round(float(np.mean([r.detector_executed for r in recs])), 2)
                                           if hasattr(recs[0], "detector_executed") else None}
                out[name + "_host_us_per_step"] = {"collect": round(1e6 * t_col / (n - 1 - PASS_START), 1),
                                                   "set_frame": round(1e6 * t_set / (n - 1 - PASS_START), 1),
                                                   "submit": round(1e6 * t_sub / (n - 1 - PASS_START), 1)}
        out[name + "_frames_per_s"] = round(float(rate), 1)
        pipe.close()
    out["note"] = ("cfg-2 loop, %d steps of the forward stream, frames uploaded from host memory every step (4 frame slots in HBM); "
                   "PCIe-inclusive, never `value`" % (n - 1 - PASS_START))
    return out


def loop_stats(records):
    """What the loop did over `records` (StepResults of all sequences of the timed steps)."""
    return {"features_in_median": int(np.median([r.n_features_in for r in records])),
            "tracked_median": int(np.median([r.n_tracked for r in records])),
            "landmarks_p3p_median": int(np.median([r.n_triangulated for r in records])),
            "inliers_median": float(np.median([r.n_inliers for r in records])),
            "candidates_median": float(np.median([r.n_candidates for r in records])),
            "ransac_iters_median": float(np.median([r.ransac_iterations for r in records])),
            "redetect_fraction_of_steps": float(np.mean([r.redetected for r in records])),
            "detector_executed_fraction_of_steps": float(np.mean([r.detector_ran for r in records])),
            "steps_finished_by_host_path": int(sum(r.recovered for r in records)),
            "host_path_reasons": {name: int(sum(1 for r in records if r.recovered and (r.reserved & bit)))
                                  for bit, name in ((1, "few_landmarks"), (2, "possibly_rejected_draw"), (4, "rule_not_done_after_hyp_samples"),
                                                    (8, "capacity"), (16, "forced"), (32, "detector_skipped"))
                                  if any(r.recovered and (r.reserved & bit) for r in records)},
            "refine_steps_median": float(np.median([r.refine_iterations for r in records]))}


def timed_leg(ctx, pipe, S, warm, steps):
    """warm untimed + `steps` timed steps of `pipe` (look-ahead, seams included); the leg's dict."""
    w = Walker(pipe, N_FRAMES)
    recs = []
    w.run(warm)
    ctx.sync()
    t0 = time.perf_counter()
    w.run(steps, on_step=lambda b, rs: recs.extend(rs))
    ctx.sync()
    dt = time.perf_counter() - t0
    st = loop_stats(recs)
    return {"frames_per_s": round(S * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
            "sequences_per_gpu": S, "passes_started_in_timed_region": w.passes,
            "redetect_fraction_of_steps": round(st["redetect_fraction_of_steps"], 4),
            "detector_executed_fraction_of_steps": round(st["detector_executed_fraction_of_steps"], 4),
            "steps_finished_by_host_path": st["steps_finished_by_host_path"], "host_path_reasons": st["host_path_reasons"],
            "inliers_median": st["inliers_median"], "landmarks_p3p_median": st["landmarks_p3p_median"],
            "ransac_iters_median": st["ransac_iters_median"]}


class _Stdout:
    """The process's real stdout kept aside for the ONE JSON line; file descriptor 1 itself is pointed at stderr, so
    that what native libraries print (RCCL's version banner at communicator creation) cannot precede or follow the line."""
    file = None

    @classmethod
    def isolate(cls):
        if cls.file is None:
            sys.stdout.flush()
            cls.file = os.fdopen(os.dup(1), "w")
            os.dup2(2, 1)

    @classmethod
    def emit(cls, obj):
        f = cls.file or sys.stdout
        f.write(json.dumps(obj) + "\n")
        f.flush()


def main():
    _Stdout.isolate()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the drop-in API frames/s leg")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the in-line detector-every-frame and 16-sequences legs (profiling runs of the headline alone)")
    ap.add_argument("--no-lookahead", dest="lookahead", action="store_false",
                    help="one blocking vo_pipeline_step per frame instead of submitting frame k+1 before "
                         "collecting frame k (vo_pipeline_submit / _collect)")
    ap.add_argument("--sequences", type=int, default=int(os.environ.get("VO_BENCH_SEQUENCES", "1")),
                    help="independent sequences per GPU advancing through the same launches (vo_pipeline_config.sequences); "
                         "the headline stays at 1, the line carries a 16-sequences leg")
    ap.add_argument("--exchange", action="store_true",
                    help="run the all-gather of {pose, landmarks} records even on one GPU (always on for --gpus > 1)")
    args = ap.parse_args()
    if CONFIG in ("cfg3", "harris"):
        if args.steps == 2000:
            args.steps, args.warmup = 150, 10
        return cfg3_main(args, "sift" if CONFIG == "cfg3" else "harris")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    S = max(1, args.sequences)
    legs = world == 1 and not args.no_legs and CONFIG == "cfg2"
    # the frames first, in worker processes, before this process touches the GPU: every sequence of every rank is its
    # own scene (its own texture seed); the 16-sequences leg gets 16 more
    from vo import synthetic
    workers = max(1, RENDER_WORKERS // max(1, world))
    t_r = time.perf_counter()
    streams = [synthetic.Stream(N_FRAMES, H, W, seed=2023 + rank * S + q) for q in range(S)]
    leg_streams = [synthetic.Stream(N_FRAMES, H, W, seed=3023 + q) for q in range(S_LEG)] if legs and S != S_LEG else []
    jobs = [(st.start + i, H, W, st.seed) for st in streams + leg_streams for i in range(N_FRAMES)]
    for (st, i), im in zip([(st, i) for st in streams + leg_streams for i in range(N_FRAMES)],
                           synthetic.render_images(jobs, workers)):
        st._img[i] = im
    render_s = time.perf_counter() - t_r

    import torch
    import torch.distributed as dist
    torch.set_num_threads(4)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    exchange = False
    want_exchange = world > 1 or args.exchange

    from vo import _native, sharding
    # torch's own (null-stream) work first, then the pipeline's streams: a stream is attached to one of the four
    # hardware queues when it first runs, and the frame loop's four streams should not share one among themselves
    cap = N_KP
    rec_len = sharding.record_length(cap)
    recs = [torch.zeros(EXCHANGE_EVERY * S * rec_len, dtype=torch.float64, device="cuda") for _ in range(2)]
    gathered = [torch.zeros(world * EXCHANGE_EVERY * S * rec_len, dtype=torch.float64, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    comp = torch.cuda.Stream(priority=-1 if os.environ.get("VO_BENCH_MAIN_PRIORITY") == "high" else 0)
    comm = torch.cuda.Stream()
    os.environ["VO_DEVICE"] = str(local)
    # (VO_BENCH_OWN_STREAM=1: the library creates the main stream itself -- under VO_STREAM_CUS on a subset of the compute units)
    ctx = _native.Context(local, stream=None if os.environ.get("VO_BENCH_OWN_STREAM") else comp.cuda_stream)
    _native.set_default_context(ctx)      # the host classes of the bootstrap run on the same context / stream
    stream = streams[0]
    states = [bootstrap_state(st) for st in streams]
    state = states[0]
    pipe = make_pipeline(ctx, streams, states, S, DETECT_MARGIN)
    n_boot = int((state.curr_frame.features.state == 2).sum())
    boot_info = getattr(state, "bootstrap_info", {})
    walker = Walker(pipe, N_FRAMES)
    # the records of EXCHANGE_EVERY frames per all-gather, double-buffered (vo.sharding.RecordExchange; covered with CPU
    # tensors and gloo by tests/test_sharding_gloo.py)
    xchg = sharding.RecordExchange(
        recs, gathered, rec_len, EXCHANGE_EVERY, S,
        post=lambda r, q, buf, off: pipe.export_state_post(r, cap, buf.data_ptr() + off * 8, seq=q),
        join=lambda: pipe.export_state_join(comm.cuda_stream),
        gather=lambda src, dst: (sharding.allgather_records(src, dst), None)[1])

    def run(n, record=False, lookahead=None):
        # Default: one frame of look-ahead, as a camera stream gives it -- frame k+1 is submitted (all of its GPU
        # work enqueued) before the record of frame k is read.  Every frame is processed in full and the results
        # are those of the blocking call (tests/test_gpu_pipeline.py).  --no-lookahead: one blocking step per frame.
        out = []
        left = [n]

        def on_step(b, rs):
            left[0] -= 1
            if exchange:
                # The record of every collected step is queued on the pipeline's stream (no host synchronisation);
                # every EXCHANGE_EVERY frames the records gathered so far go to all ranks in ONE all-gather on the
                # side stream (fewer, larger collectives: issuing one costs the host ~45 us, a third of a step).
                xchg.post(rs)
                if left[0] == 0:
                    xchg.flush()
            if record:
                out.append((b, rs))

        walker.run(n, lookahead=args.lookahead if lookahead is None else lookahead, on_step=on_step)
        return out

    def fence():
        if exchange:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    # ---- untimed: the whole first pass, blocking.  Its first frames are kept for the parity / CPU-baseline leg, all of
    # it for the comparison with the analytic ground truth ----
    ORACLE_FRAMES = 4
    first = run(ORACLE_FRAMES, record=True, lookahead=False)
    first_state = pipe.get_state()
    first_pass = first + run(N_FRAMES - 1 - PASS_START - ORACLE_FRAMES, record=True, lookahead=False)
    ctx.sync()
    # The process group comes up only now: HIP spreads streams over its hardware queues in the order they first run,
    # and the pipeline's main stream should not end up sharing a queue with RCCL's (measured in round 1).
    if want_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        torch.cuda.set_stream(comm)      # the collectives are issued from this stream; the pipeline has its own
        exchange = True

    # untimed: warmup, and one all-kernel event profile to find the dominant kernel
    run(args.warmup)
    ctx.prof_enable(-1)
    pipe.prof_reset()
    run(16)
    per_kernel = {}
    for kid in range(_native.K_COUNT):
        ms, n = pipe.prof_read(kid)
        if n:
            per_kernel[ctx.kernel_name(kid)] = (ms, n)
    ctx.prof_disable()
    dom_name = max(per_kernel, key=lambda k: per_kernel[k][0])
    dom_id = [k for k in range(_native.K_COUNT) if ctx.kernel_name(k) == dom_name][0]

    # timed region: exactly K steps, events only around the dominant kernel (every PROF_EVERY-th launch: an event
    # pair costs the stream ~5 us, bracketing every launch would slow the step it measures)
    pipe.prof_reset()
    ctx.prof_set_sampling(PROF_EVERY)
    ctx.prof_enable(dom_id)
    passes0 = walker.passes
    fence()
    t0 = time.perf_counter()
    timed = run(args.steps, record=True)
    fence()
    dt = time.perf_counter() - t0
    dom_ms, dom_n = pipe.prof_read(dom_id)
    ctx.prof_disable()
    ctx.prof_set_sampling(1)

    dt_t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt_max = float(dt_t.item())

    if rank == 0:
        every = [r for _, rs in timed for r in rs]           # all sequences' records
        res = [rs[0] for _, rs in timed]                     # sequence 0: chain stamps
        lstats = loop_stats(every)
        n_in, n_trk, n_tri = lstats["features_in_median"], lstats["tracked_median"], lstats["landmarks_p3p_median"]
        avg_us = dom_ms / max(dom_n, 1) * 1e3

        def abytes(k):                                       # one launch processes all S sequences
            v = algorithmic_bytes(k, n_in, n_trk, n_tri)
            return None if v is None else v * S

        ab = abytes(dom_name)
        roof = {"bound": "hbm", "kernel": KERNEL_LABEL.get(dom_name, dom_name), "avg_launch_us": round(avg_us, 3), "launches": dom_n,
                "algorithmic_bytes_per_launch": ab, "sequences_per_launch": S, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "traffic": pmc_traffic(dom_name) if S == 1 else None}
        if ab and avg_us > 0:
            ach = ab / (avg_us * 1e-6) / 1e9
            roof["achieved"] = round(ach, 2)
            roof["frac"] = round(ach / HBM_PEAK_GBS, 5)
        else:
            roof["achieved"] = None
            roof["frac"] = None

        def us(k):
            return per_kernel[k][0] / per_kernel[k][1] * 1e3

        # pose against the analytic ground truth of the stream over the whole first pass: the bootstrap fixes the unit of
        # length (|t| = 1 between frames 0 and 2, 1.6 m apart), positions compared after that one scale
        scale = 2 * synthetic.STEP_Z / max(np.linalg.norm(state.curr_pose[:3, 3]), 1e-12)
        gt_err = []
        for b, rs in first_pass:
            r = rs[0]
            Twc = r.pose_world_cam()
            gt = np.linalg.inv(stream.T_world_cam(0)) @ stream.T_world_cam(b)
            gt_err.append((float(np.linalg.norm(Twc[:3, :3] - gt[:3, :3])), float(np.linalg.norm(scale * Twc[:3, 3] - gt[:3, 3]))))
        # device clock (100 MHz) stamps the kernels of the chain leave in every record: where a step's time goes
        # (steps right behind a seam start from a drained pipeline: their period is the seam's, not the chain's)
        ts = np.array([[r.ts[k] for k in range(8)] for r in res], dtype=np.float64) * 1e-2          # us
        period = np.diff(ts[:, 1])
        seam = np.array([timed[i + 1][0] == PASS_START + 1 for i in range(len(timed) - 1)], dtype=bool)
        inpass = period[~seam] if (~seam).any() else period
        chain = {"tracker_start_to_regroup_start": float(np.median(ts[:, 1] - ts[:, 0])),
                 "regroup_to_hypotheses": float(np.median(ts[:, 2] - ts[:, 1])),
                 "hypotheses_to_pose": float(np.median(ts[:, 3] - ts[:, 2])),
                 "pose_to_landmarks": float(np.median(ts[:, 4] - ts[:, 3])),
                 "landmarks_to_record": float(np.median(ts[:, 5] - ts[:, 4])),
                 "record_to_next_regroup": float(np.median((ts[1:, 1] - ts[:-1, 5])[~seam])) if (~seam).any() else None,
                 "regroup_to_next_tracker_start": float(np.median((ts[1:, 0] - ts[:-1, 1])[~seam])) if (~seam).any() else None,
                 "step_period": float(np.median(inpass)),
                 "step_period_mean": float(np.mean(inpass)),
                 "step_period_percentiles_10_50_90_99": [float(v) for v in np.percentile(inpass, [10, 50, 90, 99])],
                 "seam_period": [float(v) for v in period[seam]][:8],
                 "redetect_step_period_median": (float(np.median([period[i] for i in range(len(period)) if res[i + 1].redetected and not seam[i]]))
                                                 if any(res[i + 1].redetected and not seam[i] for i in range(len(period))) else None),
                 "unit": "us, medians over the timed steps, from wall_clock64() stamps of each kernel's first work item"}
        if os.environ.get("VO_POSE_STAMPS"):
            chain["pose_kernel_replay_refine_candidates"] = [float(np.median(ts[:, 6] - ts[:, 3])), float(np.median(ts[:, 7] - ts[:, 6])),
                                                             float(np.median(ts[:, 4] - ts[:, 7]))]
        out = {
            "metric": "VO frames/sec at 1376x1241, 2k keypoints; pose err vs reference" if CONFIG != "cfg5" else
                      "VO frames/sec at 3840x2160, 8k keypoints (stress configuration)",
            "value": round(world * S * args.steps / dt_max, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%d KITTI-shaped synthetic stream, %d frames strictly forward (0.8 m/frame), per frame: "
                                   "Harris+NMS %d kp (see detector), re-detect append, KLT %d-level 15x15 of all tracks, Matches "
                                   "regroup, P3P-RANSAC %d hyps + device replay of the sequential rule, pose refinement, State "
                                   "bookkeeping, per-track-pose candidate DLT, cheirality; %d independent sequence(s) per GPU; "
                                   "detector executed on %.0f %% of the timed steps, re-detect on %.0f %%"
                                   % ("cfg-5" if CONFIG == "cfg5" else "cfg-2", W, H, N_FRAMES, N_KP, MAX_LEVEL + 1, HYP, S,
                                      100 * lstats["detector_executed_fraction_of_steps"], 100 * lstats["redetect_fraction_of_steps"]),
                       "step_contains": "all of the above, device-resident (Features/State/RANSAC never leave HBM); "
                                        "landmarks are the loop's own triangulations after a host two-view bootstrap",
                       "frames_resident": N_FRAMES, "stream": "frames %d -> %d pass after pass; seam = drain + vo_pipeline_rewind "
                                                                "to the bootstrap's state of frame %d, inside the timed region"
                                                                % (PASS_START, N_FRAMES - 1, PASS_START),
                       "passes_started_in_timed_region": walker.passes - passes0,
                       "keypoints": N_KP, "hypotheses": HYP, "hypotheses_note": "RANSAC budget %d iterations (max_iterations); %d samples solved + scored per "
                                                                       "launch (iterations without a P3P solution are not counted, "
                                                                       "ransac.py:98-101)" % (HYP, HYP_LAUNCH),
                       "sequences_per_gpu": S,
                       "frame_lookahead": 1 if args.lookahead else 0, "redetect_start_pose": REDETECT_POSE,
                       "detector": ("Harris + NMS on every frame" if DETECT_MARGIN < 0 else
                                    "Harris + NMS launched every frame, executed for a sequence whose track count, extrapolated "
                                    "by 2.5 times its last loss, is below %.2f x num_features (re-detect limit 0.80, as "
                                    "klt.py:207-230; a sequence that falls through that in one frame is finished by the host "
                                    "path)" % (0.8 + DETECT_MARGIN)),
                       "rccl_world_size": dist.get_world_size() if exchange else 1,
                       "parallelism": "sequence-sharded x%d%s%s" % (world, ", %d sequences per GPU per launch" % S if S > 1 else "", ", RCCL all-gather of the {pose, landmarks} records of %d frames every %d frames" % (EXCHANGE_EVERY, EXCHANGE_EVERY) if exchange else "")},
            "roofline": roof,
            # the image-wide (streaming) kernels against the same HBM peak, from the untimed all-kernel event pass
            # (the detector's kernels only when they run on every frame: a launch that a sequence sits out returns at once
            #  and would flatter the average; VO_BENCH_DETECT_MARGIN=-1 gives their figures, profiles/ holds that line)
            "roofline_streaming": {k: {"avg_launch_us": round(us(k), 2),
                                       "algorithmic_bytes_per_launch": abytes(k),
                                       "achieved": round(abytes(k) / (us(k) * 1e-6) / 1e9, 1),
                                       "frac": round(abytes(k) / (us(k) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
                                   for k in (("harris_response", "nms_candidates", "pyr_down") if DETECT_MARGIN < 0 else ("pyr_down",))
                                   if k in per_kernel},
            "per_kernel_us": {k: round(us(k), 2) for k in sorted(per_kernel)},
            "per_kernel_note": "HIP-event averages over 16 untimed steps with every kernel bracketed (streams overlap: a kernel's "
                               "time includes what it waits for the others)" + ("" if DETECT_MARGIN < 0 else
                               "; harris_response / nms_* average over launches most sequences sit out"),
            "chain_us": chain,
            "loop": dict(lstats, bootstrap_landmarks=n_boot),
            "pose_err_vs_ground_truth": {"rot_fro_median": float(np.median([e[0] for e in gt_err])),
                                         "rot_fro_max": float(np.max([e[0] for e in gt_err])),
                                         "trans_m_median": float(np.median([e[1] for e in gt_err])),
                                         "trans_m_last": round(gt_err[-1][1], 3),
                                         "trans_m_every_8th_frame": [round(e[1], 3) for e in gt_err[::8]],
                                         "path_m": round((N_FRAMES - 1 - PASS_START) * synthetic.STEP_Z, 1),
                                         "frames": len(gt_err),
                                         "steps_finished_by_host_path": int(sum(rs[0].recovered for _, rs in first_pass)),
                                         "note": "the whole first pass (frames %d..%d) vs the analytic poses of the synthetic stream, "
                                                 "monocular scale fixed once by the bootstrap baseline" % (PASS_START + 1, N_FRAMES - 1)},
            "setup_s": {"render_%d_frames_%d_workers" % (len(jobs), workers): round(render_s, 1)},
            "bootstrap": {"relative_pose_ms": round(1e3 * boot_info.get("relative_pose_seconds", 0.0), 3),
                          "correspondences": boot_info.get("correspondences"), "inliers": boot_info.get("inliers"),
                          "note": "triangulate_matches of frames 0 and 2 (triangulation.py:88-350): 8-point RANSAC hypotheses + "
                                  "counts, closing fit, E decomposition, four cheirality votes and the triangulation on the "
                                  "device; the sequential accept rule and the Python classes around it on the host; untimed"},
        }
        if world == 1 and not args.no_cpu_baseline:
            parity, base = oracle_leg(stream, state, [rs[0] for _, rs in first_pass[:CPU_BASELINE_FRAMES]], first_state, ORACLE_FRAMES)
            out["pose_vs_oracle"] = parity
            out["cpu_baseline"] = base
        if world == 1 and not args.no_api:
            # the API legs with every other pipeline closed (a second pipeline's streams share the device's hardware queues
            # with the first one's: beside the open headline pipeline these legs ran at half their rate)
            if pipe is not None:
                pipe.close()
                pipe = None
            _native.release_cached()         # (and the closed pipeline's masked side streams gone, not kept for a next one)
            out["api"] = api_leg(ctx)
            try:
                if S == 1:
                    out["api"]["frames_from_host_memory"] = upload_leg(ctx, stream, state)
            except Exception as e:                           # the headline must not depend on this leg
                out["api"]["frames_from_host_memory"] = {"error": repr(e)}
        if legs:
            if pipe is not None:
                pipe.close()
                pipe = None
            if DETECT_MARGIN >= 0:
                # the same loop with the detector executed on EVERY frame (what rounds 1 and early 2 timed), in the same line
                p2 = make_pipeline(ctx, streams, states, S, -1.0)
                out["detector_every_frame"] = dict(timed_leg(ctx, p2, S, 100, 1000),
                                                   note="same pipeline, same stream, detect_margin < 0: Harris + NMS executed on every "
                                                        "frame of every sequence instead of within the margin of the re-detect limit")
                p2.close()
            if S != S_LEG:
                # throughput configuration: 16 independent sequences (16 scenes) per GPU advancing through the same launches
                t_b = time.perf_counter()
                leg_states = [bootstrap_state(st) for st in leg_streams]
                out["setup_s"]["bootstrap_%d_sequences" % S_LEG] = round(time.perf_counter() - t_b, 1)
                p3 = make_pipeline(ctx, leg_streams, leg_states, S_LEG, DETECT_MARGIN)
                out["sequences_16"] = dict(timed_leg(ctx, p3, S_LEG, 30, 2 * (N_FRAMES - 1 - PASS_START)),
                                           note="16 scenes (seeds 3023..3038), each bootstrapped by itself, one pipeline: every launch "
                                                "of the loop carries all 16 sequences; two passes over the stream timed, seams included")
                p3.close()
                if DETECT_MARGIN >= 0:
                    p4 = make_pipeline(ctx, leg_streams, leg_states, S_LEG, -1.0)
                    out["sequences_16_detector_every_frame"] = timed_leg(ctx, p4, S_LEG, 30, 2 * (N_FRAMES - 1 - PASS_START))
                    p4.close()
        _Stdout.emit(out)
    if pipe is not None:
        pipe.close()
    ctx.close()
    if exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
