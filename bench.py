#!/usr/bin/env python3
"""Benchmark: VO front-end frames/s on the BASELINE.json config-2 workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

One "step" = the whole per-frame front-end on one 1376x1241 frame that is already
resident in HBM: pyramid -> KLT (3 levels, 15x15) of 2000 keypoints -> Harris
response + exact NMS (2000 keypoints) on the new frame -> P3P-RANSAC (1000
hypotheses solved + scored on the GPU, reference-exact sampler and accept rule)
-> DLT triangulation.  Each rank runs its own synthetic sequence (weak scaling,
frame streams shard at sequence granularity); with N > 1 every step all-gathers the
ranks' {pose, landmarks} records over RCCL on a side stream.  Rank 0 prints ONE
JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "visual-odometry-project_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

H, W, N_KP, HYP, WIN, MAX_LEVEL = 1241, 1376, 2000, 1000, 15, 2
N_FRAMES = 8
REFINE_ITERS = int(os.environ.get("VO_BENCH_REFINE", "20"))   # Gauss-Newton steps allowed to the pose refinement (0: off)
EXCHANGE_EVERY = int(os.environ.get("VO_BENCH_EXCHANGE_EVERY", "16"))   # frames per all-gather of {pose, landmarks} records (multi-GPU / --exchange)
PROF_EVERY = 4           # HIP-event pairs around every 4th launch of the dominant kernel in the timed region
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(kernel_name, n_tracked):
    """Compulsory HBM traffic per launch (DESIGN.md 'Kernels')."""
    px = H * W
    levels = MAX_LEVEL + 1
    table = {
        "harris_response": px * 1 + px * 8,                   # image read once, fp64 map written once
        "nms_candidates": px * 8,                             # fp64 map read once
        "nms_compact": N_KP * 12,                             # the N selected strict maxima
        "nms_round": N_KP * 4 * 2,                            # state words of the picks (list traffic is not compulsory)
        "nms_rank": N_KP * 12,
        "nms_select": N_KP * 16,                              # keypoints written
        "pyr_down": 2 * px + px // 4 + px // 16,              # one launch: frame read, bordered copy of level 0, levels 1 and 2 written
        "klt_track": N_KP * levels * ((WIN + 3) ** 2 + (WIN + 1) ** 2) + N_KP * (8 + 8 + 1 + 4),
        # (flags of all keypoints per workgroup are re-reads, not compulsory) selection of the tracked keypoints
        # + compacted arrays, then samples, poses, valid flags
        "p3p_solve": N_KP * (1 + 4 + 8 + 16 + 24) + n_tracked * (16 + 16 + 24) + HYP * (28 + 4 * 32 + 96 + 1),
        "p3p_score": n_tracked * 40 + HYP * (96 + 1 + 4 + ((n_tracked + 63) // 64) * 8),
        "dlt_triangulate": n_tracked * (16 + 16 + 24) + 192,
        "refine_pose": n_tracked * 40 + ((n_tracked + 63) // 64) * 8 + 96 + 120,   # points + mask row once, pose in / out
    }
    return table.get(kernel_name)


def pmc_traffic(kernel_name):
    """HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate runs of this
    same command, FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes); None if the
    file or the kernel is missing.  bench.py itself cannot run under the counters."""
    names = {"klt_track": "klt_track16_kernel<15, 16>", "nms_round": "nms_round_kernel<5, true>",
             "nms_candidates": "nms_candidates_kernel<5>", "harris_response": "harris_response_kernel<9>",
             "p3p_solve": "p3p_solve_kernel<true>", "p3p_score": "p3p_score_kernel", "dlt_triangulate": "dlt_kernel",
             "nms_compact": "nms_compact_kernel", "nms_rank": "nms_rank_kernel", "nms_select": "nms_finalize_kernel",
             "track_gather": "gather_tracks_kernel", "refine_pose": "refine_pose_kernel", "pyr_down": "pyramid3_kernel"}
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")) as f:
            table = json.load(f)
        return int(table[names[kernel_name]]["hbm_bytes_corrected"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(stream, frames=3):
    """The oracle (CPU restatement of the reference path) timed on this host, 1 thread."""
    from oracle import dlt_np, harris_np, native, ransac_np
    K = stream.K
    kp = harris_np.nms_keypoints_fast(harris_np.harris_scores(stream.image(0), 9, 0.09), N_KP, 5)[:, :, 0]
    rs = ransac_np.Ransac(4, np.arange(4), None, None, 1.0, 0.9, 0.99, 1000, adaptive=True, p3p=True)
    order = stream.order(frames)
    t0 = time.perf_counter()
    for a, b in zip(order[:-1], order[1:]):
        out, status, err = native.klt_track(stream.image(a), stream.image(b), kp.astype(np.float32), win=WIN,
                                            max_level=MAX_LEVEL)
        keep = status.astype(bool) & (err < 100.0)
        p_c, n_c = kp[keep], out[keep].astype(np.float64)
        z = stream.depth(a)[p_c[:, 1].astype(int), p_c[:, 0].astype(int)].astype(np.float64)
        T = stream.T_world_cam(a)
        xc, yc = (p_c[:, 0] - K[0, 2]) / K[0, 0] * z, (p_c[:, 1] - K[1, 2]) / K[1, 1] * z
        land = np.stack([T[r, 0] * xc + T[r, 1] * yc + T[r, 2] * z + T[r, 3] for r in range(3)], axis=1)
        # the reference's own NMS loop: 2*N full-map argmax passes (harris.py:148-152)
        kp = harris_np.nms_keypoints(harris_np.harris_scores(stream.image(b), 9, 0.09), N_KP, 5)[:, :, 0]
        rs.model_fn = lambda idx: native.p3p_solve(land[np.asarray(idx).reshape(-1)], n_c[np.asarray(idx).reshape(-1)], K)
        rs.error_fn = lambda m, pop: native.reproj_errors(land, n_c, K, m[0], m[1])
        (R, t), inl = rs.find_best_model(np.arange(len(land)))
        dlt_np.linear_triangulation(p_c, n_c, K @ np.linalg.inv(T)[:3], K @ np.hstack([R, t[:, None]]))
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames of the same 1376x1241 stream through oracle/ (NumPy Harris + the reference's "
                      "2N-argmax NMS loop, C KLT/P3P, NumPy RANSAC/DLT), single thread" % frames}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lookahead", dest="lookahead", action="store_false",
                    help="one blocking vo_pipeline_step per frame instead of submitting frame k+1 before "
                         "collecting frame k (vo_pipeline_submit / _collect)")
    ap.add_argument("--exchange", action="store_true",
                    help="run the per-frame all-gather of {pose, landmarks} records even on one GPU "
                         "(always on for --gpus > 1); rehearses the multi-GPU step on a single device")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node=%d" % args.gpus
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    exchange = False                     # switched on below, once the pipeline's streams exist and have run
    want_exchange = world > 1 or args.exchange

    from vo import _native, sharding, synthetic
    comp = torch.cuda.Stream()
    comm = torch.cuda.Stream()
    ctx = _native.Context(local, stream=comp.cuda_stream)
    stream = synthetic.Stream(N_FRAMES, H, W, seed=2023 + rank)
    pipe = _native.Pipeline(ctx, H, W, N_FRAMES, stream.K, n_keypoints=N_KP, klt_win=WIN, klt_max_level=MAX_LEVEL,
                            hyp=HYP, p3p_threshold=1.0, outlier_ratio=0.9, confidence=0.99, max_iterations=1000,
                            refine_iters=REFINE_ITERS)
    for i in range(N_FRAMES):
        pipe.set_frame(i, stream.image(i), stream.depth(i), stream.T_world_cam(i))

    cap = N_KP
    rec_len = sharding.record_length(cap)
    recs = [torch.zeros(EXCHANGE_EVERY * rec_len, dtype=torch.float64, device="cuda") for _ in range(2)]
    gathered = [torch.zeros(world * EXCHANGE_EVERY * rec_len, dtype=torch.float64, device="cuda") for _ in range(2)]
    batch_fill, batch_buf = 0, 0

    order = stream.order(args.warmup + args.steps + 64)
    pipe.prime(order[0])
    pos = 0
    stats = {"tracked": [], "inliers": [], "rot_err": [], "trans_err": [], "iters": [], "tri_err": [],
             "rot_err_ref": [], "trans_err_ref": [], "ref_iters": []}

    def run(n, record=False):
        # Default: one frame of look-ahead, as a camera stream gives it -- frame k+1 is submitted (all of
        # its GPU work enqueued) before the pose of frame k is collected, so the host's share of a step
        # (launches, sequential RANSAC replay) overlaps the GPU's.  Every frame is processed in full
        # and the results are those of the blocking call (tests/test_gpu_pipeline.py).
        # --no-lookahead: the reference's order, one blocking vo_pipeline_step per frame.
        nonlocal pos
        if args.lookahead:
            pipe.submit(order[pos], order[pos + 1])
        for k in range(n):
            a, b = order[pos], order[pos + 1]
            pos += 1
            if args.lookahead:
                if k + 1 < n:
                    pipe.submit(order[pos], order[pos + 1])
                r = pipe.collect()
            else:
                r = pipe.step(a, b)
            if exchange:
                # The record of every collected step is written behind its DLT on the pipeline's own stream
                # (no host synchronisation); every EXCHANGE_EVERY frames the records gathered so far go to
                # all ranks in ONE all-gather on the side stream (fewer, larger collectives: the per-call
                # host cost of a collective is ~45 us, a third of a step).
                nonlocal batch_fill, batch_buf
                pipe.export_state_post(r, cap, recs[batch_buf].data_ptr() + batch_fill * rec_len * 8)
                batch_fill += 1
                if batch_fill == EXCHANGE_EVERY or k == n - 1:
                    pipe.export_state_join(comm.cuda_stream)
                    sharding.allgather_records(recs[batch_buf], gathered[batch_buf])
                    batch_buf ^= 1
                    batch_fill = 0
            if record:
                Tcw = np.linalg.inv(stream.T_world_cam(b))
                R, t = np.array(r.R).reshape(3, 3), np.array(r.t)
                Rr, tr = np.array(r.R_refined).reshape(3, 3), np.array(r.t_refined)
                stats["rot_err_ref"].append(float(np.linalg.norm(Rr - Tcw[:3, :3])))
                stats["trans_err_ref"].append(float(np.linalg.norm(tr - Tcw[:3, 3])))
                stats["ref_iters"].append(r.refine_iterations)
                stats["tracked"].append(r.n_tracked)
                stats["inliers"].append(r.n_inliers)
                stats["iters"].append(r.ransac_iterations)
                stats["rot_err"].append(float(np.linalg.norm(R - Tcw[:3, :3])))
                stats["trans_err"].append(float(np.linalg.norm(t - Tcw[:3, 3])))

    def fence():
        if exchange:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    # The process group comes up only now.  HIP spreads streams over four hardware queues in the order they
    # first run; the pipeline's tracking stream must not end up sharing a queue with a detection stream
    # (kernels of one hardware queue run in order: measured 8.5k -> 6k frames/s when RCCL's streams were
    # created first and shifted that assignment), so the pipeline runs a few steps before RCCL exists.
    run(4)
    ctx.sync()
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
    run(8)
    per_kernel = {}
    for kid in range(_native.K_COUNT):
        ms, n = pipe.prof_read(kid)
        if n:
            per_kernel[ctx.kernel_name(kid)] = (ms, n)
    ctx.prof_disable()
    dom_name = max(per_kernel, key=lambda k: per_kernel[k][0])
    dom_id = [k for k in range(_native.K_COUNT) if ctx.kernel_name(k) == dom_name][0]

    # timed region: exactly K steps, events only around the dominant kernel
    pipe.prof_reset()
    # (every PROF_EVERY-th launch is bracketed: an event pair costs the stream ~5 us, and bracketing
    #  all of them made the step 4 % slower than it is)
    ctx.prof_set_sampling(PROF_EVERY)
    ctx.prof_enable(dom_id)
    fence()
    t0 = time.perf_counter()
    run(args.steps, record=False)
    fence()
    dt = time.perf_counter() - t0
    dom_ms, dom_n = pipe.prof_read(dom_id)
    ctx.prof_disable()
    ctx.prof_set_sampling(1)

    # untimed: accuracy against the analytic ground truth of the stream
    run(16, record=True)
    fence()
    last = pipe.step(order[pos], order[pos + 1])
    ntr = last.n_tracked

    dt_t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt_max = float(dt_t.item())

    if rank == 0:
        avg_us = dom_ms / max(dom_n, 1) * 1e3
        ab = algorithmic_bytes(dom_name, ntr)
        roof = {"bound": "hbm", "kernel": dom_name, "avg_launch_us": round(avg_us, 3), "launches": dom_n,
                "algorithmic_bytes_per_launch": ab, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "traffic": pmc_traffic(dom_name)}
        if ab and avg_us > 0:
            ach = ab / (avg_us * 1e-6) / 1e9
            roof["achieved"] = round(ach, 2)
            roof["frac"] = round(ach / HBM_PEAK_GBS, 5)
        else:
            roof["achieved"] = None
            roof["frac"] = None
        out = {
            "metric": "VO frames/sec at 1376x1241, 2k keypoints; pose err vs reference",
            "value": round(world * args.steps / dt_max, 2),
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
            "config": {"workload": "cfg-2: 1376x1241 KITTI-shaped synthetic stream, Harris+NMS 2000 kp -> KLT 3-level "
                                   "15x15 -> P3P-RANSAC 1000 hyps -> DLT; one independent sequence per GPU",
                       "frames_resident": N_FRAMES, "keypoints": N_KP, "hypotheses": HYP,
                       "frame_lookahead": 1 if args.lookahead else 0,
                       "parallelism": "sequence-sharded x%d%s" % (world, ", RCCL all-gather of the {pose, landmarks} records of %d frames every %d frames" % (EXCHANGE_EVERY, EXCHANGE_EVERY) if exchange else "")},
            "roofline": roof,
            # the image-wide (streaming) kernels against the same HBM peak, from the untimed all-kernel event pass
            "roofline_streaming": {k: {"avg_launch_us": round(per_kernel[k][0] / per_kernel[k][1] * 1e3, 2),
                                       "algorithmic_bytes_per_launch": algorithmic_bytes(k, ntr),
                                       "achieved": round(algorithmic_bytes(k, ntr) / (per_kernel[k][0] / per_kernel[k][1] * 1e-3) / 1e9, 1),
                                       "frac": round(algorithmic_bytes(k, ntr) / (per_kernel[k][0] / per_kernel[k][1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                   for k in ("harris_response", "nms_candidates", "pyr_down") if k in per_kernel},
            "per_kernel_us": {k: round(v[0] / v[1] * 1e3, 2) for k, v in sorted(per_kernel.items())},
            "pose_err": {"rot_fro_median": float(np.median(stats["rot_err"])), "trans_m_median": float(np.median(stats["trans_err"])),
                         "tracked_median": float(np.median(stats["tracked"])), "inliers_median": float(np.median(stats["inliers"])),
                         "ransac_iters_median": float(np.median(stats["iters"])),
                         "refined_rot_fro_median": float(np.median(stats["rot_err_ref"])),
                         "refined_trans_m_median": float(np.median(stats["trans_err_ref"])),
                         "refine_steps_median": float(np.median(stats["ref_iters"])),
                         "note": "vs analytic ground truth of the synthetic stream; rot/trans: best RANSAC hypothesis, "
                                 "refined_*: after the Gauss-Newton refinement over its inliers (inside the timed step "
                                 "when refine_steps >= 0)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(stream)
        print(json.dumps(out), flush=True)
    pipe.close()
    ctx.close()
    if exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
