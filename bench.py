#!/usr/bin/env python3
"""Benchmark: VO frames/s on the BASELINE.json config-2 workload, through the device-resident frame loop.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU under torch.distributed.run; started that way by the driver, or by this script
   itself when it is called directly with --gpus N)

One "step" = one frame of the reference's steady-state loop (src/main.py:248-286, KLT tracker mode with the
Harris detector) on a 1376x1241 frame that is already resident in HBM, everything on the GPU:
  pyramid(next) | Harris response + exact greedy NMS (2000 keypoints) on next, for a sequence whose track count is
  within 2 % (+ four times its last loss) of the re-detect limit (the reference runs its detector only below the limit, klt.py:207-230;
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
N_FRAMES = 8
REFINE_ITERS = int(os.environ.get("VO_BENCH_REFINE", "20"))   # Gauss-Newton steps allowed to the pose refinement (0: off)
EXCHANGE_EVERY = int(os.environ.get("VO_BENCH_EXCHANGE_EVERY", "16"))   # frames per all-gather of {pose, landmarks} records
REDETECT_POSE = os.environ.get("VO_BENCH_REDETECT_POSE", "current")   # see vo_pipeline_config.redetect_start_pose
DETECT_MARGIN = float(os.environ.get("VO_BENCH_DETECT_MARGIN", "0.02"))   # see vo_pipeline_config.detect_margin (< 0: every frame)
PROF_EVERY = 4           # HIP-event pairs around every 4th launch of the dominant kernel in the timed region
HBM_PEAK_GBS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
PROFILE_TAG = "r02"


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
        "p3p_solve": HYP * (28 + 4 * 40 + 96 + 1) + n_tri * 40 + HYP * (4 + ((n_tri + 63) // 64) * 8),   # solve + score in one launch
        "p3p_score": n_tri * 40 + HYP * (96 + 1 + 4 + ((n_tri + 63) // 64) * 8),
        "ransac_replay": HYP * 5 + 96 + ((n_tri + 63) // 64) * 16,
        "refine_pose": n_tri * 40 + ((n_tri + 63) // 64) * 8 + 96 + 120,
        "state_candidates": n_tracked * (16 + 1 + 16 + 96 + 1) + ((n_tri + 63) // 64) * 8,
        "state_landmarks": n_tracked * (1 + 1 + 24 + 16 + 96 + 24),
    }
    # the frame loop's pose kernel (kernel id "refine_pose"): replay + refinement + candidates + landmark stage in one launch
    table["refine_pose"] = table["ransac_replay"] + table["refine_pose"] + table["state_candidates"] + table["state_landmarks"]
    return table.get(kernel_name)


KERNEL_LABEL = {"refine_pose": "frame_pose (RANSAC replay + pose refinement + State bookkeeping + candidate DLT + record, one workgroup)",
                "p3p_solve": "p3p_hyp (P3P solve + inlier counts of all hypotheses)"}
ROCPROF_NAMES = {"klt_track": "klt_track16_kernel<15, 16>", "nms_round": "nms_round_kernel<5, true>",
                 "nms_candidates": "nms_candidates_kernel<5>", "harris_response": "harris_response_kernel<9>",
                 "p3p_solve": "p3p_hyp_kernel<8>", "p3p_score": "p3p_score_kernel", "nms_compact": "nms_compact_kernel",
                 "nms_rank": "nms_rank_kernel", "nms_select": "nms_finalize_kernel", "refine_pose": "frame_pose_kernel",
                 "pyr_down": "pyramid3_tiled_kernel<32, 16>", "state_regroup": "state_regroup_klt_kernel",
                 "ransac_replay": "ransac_replay_kernel", "state_candidates": "state_candidates_kernel",
                 "state_landmarks": "state_landmarks_kernel"}


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


def walk(start, n, steps):
    """Frame indices from `start`, back and forth over 0..n-1 (consecutive frames are always neighbours)."""
    idx, d, out = start, 1, [start]
    for _ in range(steps):
        if idx + d < 0 or idx + d >= n:
            d = -d
        idx += d
        out.append(idx)
    return out


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


def oracle_leg(stream, state, gpu_results, gpu_state, frames):
    """The CPU oracle of the same loop (tests/pipeline_oracle.py: pinned bookkeeping classes + CPU oracles) on the
    first `frames` frames: (i) parity of the GPU results against it, (ii) its time = the CPU baseline ("port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import harris_np
    from pipeline_oracle import OracleLoop
    orc = OracleLoop(stream, N_KP, WIN, MAX_LEVEL, refine_iters=REFINE_ITERS, redetect_start_pose=REDETECT_POSE)
    orc.set_state(2, state.curr_frame.features, state.curr_pose, state.prev_pose)
    order = walk(2, stream.n, frames)
    dR = dt = 0.0
    exact = True
    t0 = time.perf_counter()
    refs = []
    for b in order[1:]:
        refs.append(orc.step(b))
        # the GPU step also runs the detector on every new frame (the keypoints the next step may append)
        harris_np.nms_keypoints_fast(harris_np.harris_scores(stream.image(b), 9, 0.09), N_KP, 5)
    cpu_s = time.perf_counter() - t0
    for ref, r in zip(refs, gpu_results):
        Rr, tr = np.array(r.R_refined).reshape(3, 3), np.array(r.t_refined)
        dR = max(dR, float(np.abs(Rr - ref["R_ref"]).max()))
        dt = max(dt, float(np.abs(tr - ref["t_ref"]).max() / max(1.0, np.linalg.norm(ref["t_ref"]))))
        exact &= (r.n_tracked, r.n_triangulated, r.n_inliers, r.draws_consumed, r.ransac_iterations, r.n_candidates,
                  r.n_landmarks) == (ref["n_tracked"], ref["n_tri"], ref["n_inliers"], ref["draws"], ref["iters"],
                                     ref["n_cand"], ref["n_landmarks"])
    f = refs[-1]["features"]
    exact &= bool(np.array_equal(gpu_state["keypoints"], f.keypoints.astype(np.float64)) and
                  np.array_equal(gpu_state["state"], f.state) and
                  np.array_equal(gpu_state["candidate_mask"], f.candidate_mask) and
                  np.array_equal(gpu_state["tracks"], f.tracks, equal_nan=True))
    with np.errstate(invalid="ignore"):
        d = np.abs(gpu_state["landmarks"] - f.landmarks)[:, :, 0].max(axis=1)
        lm = float(np.nanmax(d / np.maximum(1.0, np.linalg.norm(f.landmarks[:, :, 0], axis=1)))) if f.length else 0.0
    parity = {"frames": frames, "max_abs_dR": dR, "max_rel_dt": dt, "max_rel_dlandmark": lm,
              "counts_masks_keypoints_states_tracks_exact": bool(exact),
              "note": "refined pose / landmarks vs the CPU oracle of the same loop on the same frames (tolerance of the "
                      "metric: 1e-4 rel.); integer and index results must be identical"}
    base = {"value": frames / cpu_s, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames of the same stream through the CPU oracle of the loop (tests/pipeline_oracle.py: "
                      "C KLT / P3P, NumPy RANSAC / refinement / DLT / bookkeeping) plus NumPy Harris + the oracle's fast "
                      "exact NMS walk per frame, single thread" % frames}
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
                       "settings (1.25 px, confidence 0.9999, up to 10000 iterations)")
    except Exception as e:                                   # the headline must not depend on this leg
        out["error"] = repr(e)
    return out


def cfg3_main(args):
    """VO_BENCH_CONFIG=cfg3: BASELINE.json configs[2] -- per frame SIFT detect + describe (cap 2000) on a 1376x1241
    frame and brute-force L2 2-NN + ratio + uniqueness matching against the previous frame's descriptors
    (src/vo/features/sift.py:23-56), through the C ABI's host entry points (vo_sift / vo_match_knn2_ratio: the frame
    goes over PCIe once, keypoints and descriptors come back).  A secondary line: the default run stays cfg-2."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    from vo import _native, synthetic
    ctx = _native.Context(0)
    stream = synthetic.Stream(N_FRAMES, H, W)
    order = walk(0, N_FRAMES, args.warmup + args.steps + 8)
    state = {"desc": ctx.sift(stream.image(order[0]), cap=2000)[1], "pos": 0}
    counts = []

    def run(n):
        for _ in range(n):
            state["pos"] += 1
            _, d = ctx.sift(stream.image(order[state["pos"]]), cap=2000)
            pairs = ctx.match_knn2_ratio(state["desc"], d, 0.8)
            counts.append((len(d), len(pairs)))
            state["desc"] = d

    run(args.warmup)
    ctx.prof_enable(-1)
    ctx.prof_reset()
    run(8)
    per = {}
    for kid in range(_native.K_COUNT):
        ms, n = ctx.prof_read(kid)
        if n:
            per[ctx.kernel_name(kid)] = ms / n * 1e3
    ctx.prof_disable()
    ctx.sync()
    t0 = time.perf_counter()
    run(args.steps)
    ctx.sync()
    dt = time.perf_counter() - t0
    px = H * W
    # scale space = base 4px f32; per octave 6 Gaussian + 5 DoG writes and equal reads at 4px / 4^o
    ss_bytes = (4 * px * 4) * (4.0 / 3.0) * 22
    ss_us = per.get("sift_scale_space")
    n_desc = int(np.median([c[0] for c in counts]))
    flops = 2.0 * n_desc * n_desc * 128
    m_us = per.get("match_knn2")
    out = {"metric": "VO frames/sec at 1376x1241, 2k keypoints; pose err vs reference", "value": round(args.steps / dt, 2),
           "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "cfg-3: 1376x1241 synthetic stream, per frame SIFT detect + describe (2000 strongest) and "
                                  "brute-force L2 2-NN + 0.8 ratio + uniqueness against the previous frame, host entry "
                                  "points (PCIe-inclusive: 1.7 MB up, ~1 MB of keypoints / descriptors down per frame)",
                      "keypoints": n_desc, "matches_median": int(np.median([c[1] for c in counts]))},
           "roofline": {"bound": "hbm", "kernel": "sift_scale_space (upsample, fused row+column blurs, decimations; the octaves' last two layers and extrema searches on a second stream, inside the same bracket)",
                        "avg_launch_us": None if ss_us is None else round(ss_us, 1), "algorithmic_bytes_per_launch": int(ss_bytes),
                        "achieved": None if not ss_us else round(ss_bytes / (ss_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": None if not ss_us else round(ss_bytes / (ss_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                        "traffic": None},
           # dense i8 MFMA peak: twice the bf16 rate (MI355X_MICROARCH.md, matrix cores) = 2 x 2.5 POP/s
           "matcher": {"bound": "mfma", "kernel": "match_knn2 (v_mfma_i32_32x32x32_i8 on bytes offset by 128, exact integer distances)",
                       "avg_launch_us": None if m_us is None else round(m_us, 1), "ops": flops,
                       "achieved": None if not m_us else round(flops / (m_us * 1e-6) / 1e12, 2), "peak": 5000.0, "unit": "TOP/s",
                       "frac": None if not m_us else round(flops / (m_us * 1e-6) / 1e12 / 5000.0, 5),
                       "note": "2000 x 2000 x 128: 1 GOP, a launch of ~500 workgroups -- latency-bound, not MFMA-bound"},
           "per_kernel_us": {k: round(v, 1) for k, v in sorted(per.items())}}
    print(json.dumps(out), flush=True)
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
    raise SystemExit(subprocess.run(cmd).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the drop-in API frames/s leg")
    ap.add_argument("--no-lookahead", dest="lookahead", action="store_false",
                    help="one blocking vo_pipeline_step per frame instead of submitting frame k+1 before "
                         "collecting frame k (vo_pipeline_submit / _collect)")
    ap.add_argument("--sequences", type=int, default=int(os.environ.get("VO_BENCH_SEQUENCES", "1")),
                    help="independent sequences per GPU advancing through the same launches (vo_pipeline_config.sequences); "
                         "the headline stays at 1, profiles/ holds the lines for 4 and 16")
    ap.add_argument("--exchange", action="store_true",
                    help="run the all-gather of {pose, landmarks} records even on one GPU (always on for --gpus > 1)")
    args = ap.parse_args()
    if CONFIG == "cfg3":
        if args.steps == 2000:
            args.steps, args.warmup = 60, 5
        return cfg3_main(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    torch.set_num_threads(4)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    exchange = False
    want_exchange = world > 1 or args.exchange

    from vo import _native, sharding, synthetic
    # torch's own (null-stream) work first, then the pipeline's streams: a stream is attached to one of the four
    # hardware queues when it first runs, and the frame loop's four streams should not share one among themselves
    cap = N_KP
    S = max(1, args.sequences)
    rec_len = sharding.record_length(cap)
    recs = [torch.zeros(EXCHANGE_EVERY * S * rec_len, dtype=torch.float64, device="cuda") for _ in range(2)]
    gathered = [torch.zeros(world * EXCHANGE_EVERY * S * rec_len, dtype=torch.float64, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    comp = torch.cuda.Stream()
    comm = torch.cuda.Stream()
    os.environ["VO_DEVICE"] = str(local)
    ctx = _native.Context(local, stream=comp.cuda_stream)
    _native.set_default_context(ctx)      # the host classes of the bootstrap run on the same context / stream
    # every sequence of every rank is its own scene (its own texture seed), bootstrapped by itself
    streams = [synthetic.Stream(N_FRAMES, H, W, seed=2023 + rank * S + q) for q in range(S)]
    stream = streams[0]
    pipe = _native.Pipeline(ctx, H, W, N_FRAMES, stream.K, n_keypoints=N_KP, klt_win=WIN, klt_max_level=MAX_LEVEL,
                            hyp=HYP, p3p_threshold=1.0, outlier_ratio=0.9, confidence=0.99, max_iterations=1000,
                            refine_iters=REFINE_ITERS, redetect_start_pose=REDETECT_POSE, sequences=S,
                            detect_margin=DETECT_MARGIN, debug_never_detect=int(os.environ.get("VO_BENCH_NEVER_DETECT", "0")))
    states = [bootstrap_state(st) for st in streams]
    state = states[0]
    for q in range(S):
        for i in range(N_FRAMES):
            pipe.set_frame(i, streams[q].image(i), seq=q)
        pipe.set_state(2, states[q].curr_frame.features, states[q].curr_pose, states[q].prev_pose, num_features=N_KP, seq=q)
    n_boot = int((state.curr_frame.features.state == 2).sum())

    batch_fill, batch_buf = 0, 0

    order = walk(2, N_FRAMES, args.warmup + args.steps + 96)
    pos = 0
    log = []

    def run(n, record=False, lookahead=None):
        # Default: one frame of look-ahead, as a camera stream gives it -- frame k+1 is submitted (all of its GPU
        # work enqueued) before the record of frame k is read.  Every frame is processed in full and the results
        # are those of the blocking call (tests/test_gpu_pipeline.py).  --no-lookahead: one blocking step per frame.
        nonlocal pos, batch_fill, batch_buf
        la = args.lookahead if lookahead is None else lookahead
        out = []
        if la:
            pipe.submit(order[pos], order[pos + 1])
        for k in range(n):
            a, b = order[pos], order[pos + 1]
            pos += 1
            if la:
                if k + 1 < n:
                    pipe.submit(order[pos], order[pos + 1])
                rs = pipe.collect_all()
            else:
                pipe.submit(a, b)
                rs = pipe.collect_all()
            if exchange:
                # The record of every collected step is queued on the pipeline's stream (no host synchronisation);
                # every EXCHANGE_EVERY frames the records gathered so far go to all ranks in ONE all-gather on the
                # side stream (fewer, larger collectives: issuing one costs the host ~45 us, a third of a step).
                for q in range(S):
                    pipe.export_state_post(rs[q], cap, recs[batch_buf].data_ptr() + batch_fill * rec_len * 8, seq=q)
                    batch_fill += 1
                if batch_fill == EXCHANGE_EVERY * S or k == n - 1:
                    pipe.export_state_join(comm.cuda_stream)
                    sharding.allgather_records(recs[batch_buf], gathered[batch_buf])
                    batch_buf ^= 1
                    batch_fill = 0
            if record:
                out.append((b, rs))
        return out

    def fence():
        if exchange:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    # ---- untimed: the first frames, blocking, kept for the parity / CPU-baseline leg ----
    ORACLE_FRAMES = 4
    first = run(ORACLE_FRAMES, record=True, lookahead=False)
    first_state = pipe.get_state()
    early = first + run(12, record=True, lookahead=False)     # ground-truth comparison: the frames right after the bootstrap
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
    prof_steps = run(16, record=True)
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
        res = [rs[0] for _, rs in timed]                     # sequence 0: chain stamps, ground truth, oracle
        n_in = int(np.median([r.n_features_in for r in every]))
        n_trk = int(np.median([r.n_tracked for r in every]))
        n_tri = int(np.median([r.n_triangulated for r in every]))
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

        # pose against the analytic ground truth of the stream: the bootstrap fixes the unit of length (|t| = 1 between
        # frames 0 and 2, 1.6 m apart), positions compared after that one scale
        scale = 2 * synthetic.STEP_Z / max(np.linalg.norm(state.curr_pose[:3, 3]), 1e-12)
        gt_err = []
        for b, rs in early:
            r = rs[0]
            Twc = r.pose_world_cam()
            gt = np.linalg.inv(stream.T_world_cam(0)) @ stream.T_world_cam(b)
            gt_err.append((float(np.linalg.norm(Twc[:3, :3] - gt[:3, :3])), float(np.linalg.norm(scale * Twc[:3, 3] - gt[:3, 3]))))
        # device clock (100 MHz) stamps the kernels of the chain leave in every record: where a step's time goes
        ts = np.array([[r.ts[k] for k in range(8)] for r in res], dtype=np.float64) * 1e-2          # us
        chain = {"tracker_start_to_regroup_start": float(np.median(ts[:, 1] - ts[:, 0])),
                 "regroup_to_hypotheses": float(np.median(ts[:, 2] - ts[:, 1])),
                 "hypotheses_to_pose": float(np.median(ts[:, 3] - ts[:, 2])),
                 "pose_to_landmarks": float(np.median(ts[:, 4] - ts[:, 3])),
                 "pose_kernel_replay_refine_candidates": [float(np.median(ts[:, 6] - ts[:, 3])), float(np.median(ts[:, 7] - ts[:, 6])),
                                                          float(np.median(ts[:, 4] - ts[:, 7]))],
                 "landmarks_to_record": float(np.median(ts[:, 5] - ts[:, 4])),
                 "record_to_next_regroup": float(np.median(ts[1:, 1] - ts[:-1, 5])),
                 "regroup_to_next_tracker_start": float(np.median(ts[1:, 0] - ts[:-1, 1])),
                 "step_period": float(np.median(np.diff(ts[:, 1]))),
                 "step_period_mean": float(np.mean(np.diff(ts[:, 1]))),
                 "step_period_percentiles_10_50_90_99": [float(v) for v in np.percentile(np.diff(ts[:, 1]), [10, 50, 90, 99])],
                 "slow_steps": [{"i": int(i), "period": float(v), "redetected": [int(res[i].redetected), int(res[i + 1].redetected)],
                                 "n_in": [int(res[i].n_features_in), int(res[i + 1].n_features_in)],
                                 "stages_i": [float(x) for x in np.diff(ts[i, :6])], "stages_i1": [float(x) for x in np.diff(ts[i + 1, :6])],
                                 "rec_to_regroup": float(ts[i + 1, 1] - ts[i, 5])}
                                for i, v in enumerate(np.diff(ts[:, 1])) if v > 400.0][:10],
                 "record_to_next_regroup_percentiles_10_50_90_99": [float(v) for v in np.percentile(ts[1:, 1] - ts[:-1, 5], [10, 50, 90, 99])],
                 "unit": "us, medians over the timed steps, from wall_clock64() stamps of each kernel's first work item"}
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
            "config": {"workload": "%s: %dx%d KITTI-shaped synthetic stream, per frame: Harris+NMS %d kp (see detector), re-detect "
                                   "append, KLT %d-level 15x15 of all tracks, Matches regroup, P3P-RANSAC %d hyps + device "
                                   "replay of the sequential rule, pose refinement, State bookkeeping, per-track-pose "
                                   "candidate DLT, cheirality; %d independent sequence(s) per GPU"
                                   % ("cfg-5" if CONFIG == "cfg5" else "cfg-2", W, H, N_KP, MAX_LEVEL + 1, HYP, S),
                       "step_contains": "all of the above, device-resident (Features/State/RANSAC never leave HBM); "
                                        "landmarks are the loop's own triangulations after a host two-view bootstrap",
                       "frames_resident": N_FRAMES, "keypoints": N_KP, "hypotheses": HYP, "sequences_per_gpu": S,
                       "frame_lookahead": 1 if args.lookahead else 0, "redetect_start_pose": REDETECT_POSE,
                       "detector": ("Harris + NMS on every frame" if DETECT_MARGIN < 0 else
                                    "Harris + NMS launched every frame, executed for a sequence whose track count, extrapolated "
                                    "by four times its last loss, is below %.2f x num_features (re-detect limit 0.80, as "
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
            "loop": {"features_in_median": n_in, "tracked_median": n_trk, "landmarks_p3p_median": n_tri,
                     "inliers_median": float(np.median([r.n_inliers for r in every])),
                     "candidates_median": float(np.median([r.n_candidates for r in every])),
                     "ransac_iters_median": float(np.median([r.ransac_iterations for r in every])),
                     "redetect_fraction_of_steps": float(np.mean([r.redetected for r in every])),
                     "detector_executed_fraction_of_steps": float(np.mean([r.detector_ran for r in every])),
                     "steps_finished_by_host_path": int(sum(r.recovered for r in every)),
                     "refine_steps_median": float(np.median([r.refine_iterations for r in every])),
                     "bootstrap_landmarks": n_boot},
            "pose_err_vs_ground_truth": {"rot_fro_median": float(np.median([e[0] for e in gt_err])),
                                         "trans_m_median": float(np.median([e[1] for e in gt_err])),
                                         "trans_m_per_frame": [round(e[1], 3) for e in gt_err],
                                         "frames": len(gt_err),
                                         "note": "first frames of the run vs the analytic poses of the synthetic stream, "
                                                 "monocular scale fixed once by the bootstrap baseline"},
        }
        if world == 1 and not args.no_cpu_baseline:
            parity, base = oracle_leg(stream, state, [rs[0] for _, rs in first], first_state, ORACLE_FRAMES)
            out["pose_vs_oracle"] = parity
            out["cpu_baseline"] = base
        if world == 1 and not args.no_api:
            out["api"] = api_leg(ctx)
        if world == 1 and DETECT_MARGIN >= 0 and not args.no_api:
            # the same loop with the detector executed on EVERY frame (what rounds 1 and early 2 timed), in the same line
            pipe.close()
            pipe2 = _native.Pipeline(ctx, H, W, N_FRAMES, stream.K, n_keypoints=N_KP, klt_win=WIN, klt_max_level=MAX_LEVEL,
                                     hyp=HYP, p3p_threshold=1.0, outlier_ratio=0.9, confidence=0.99, max_iterations=1000,
                                     refine_iters=REFINE_ITERS, redetect_start_pose=REDETECT_POSE, sequences=S, detect_margin=-1.0)
            for q in range(S):
                for i in range(N_FRAMES):
                    pipe2.set_frame(i, streams[q].image(i), seq=q)
                pipe2.set_state(2, states[q].curr_frame.features, states[q].curr_pose, states[q].prev_pose, num_features=N_KP,
                                seq=q)
            order2 = walk(2, N_FRAMES, 1400)

            def run2(p0, n):
                pipe2.submit(order2[p0], order2[p0 + 1])
                for k in range(n):
                    if k + 1 < n:
                        pipe2.submit(order2[p0 + k + 1], order2[p0 + k + 2])
                    pipe2.collect_all()

            run2(0, 200)
            ctx.sync()
            t2 = time.perf_counter()
            run2(200, 1000)
            ctx.sync()
            out["detector_every_frame"] = {"frames_per_s": round(S * 1000 / (time.perf_counter() - t2), 1), "steps": 1000,
                                           "note": "same pipeline, same stream, detect_margin < 0: Harris + NMS executed on every "
                                                   "frame of every sequence instead of within the margin of the re-detect limit"}
            pipe2.close()
        print(json.dumps(out), flush=True)
    pipe.close()
    ctx.close()
    if exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
