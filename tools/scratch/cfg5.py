"""Scratch: BASELINE.json configs[4] (3840x2160, 8000 keypoints, 4-level pyramid, 4000 hypotheses) through the
resident pipeline -- frames/s with one frame of look-ahead and the event-timed kernel durations.  Not the
bench line (bench.py measures configs[1]); the numbers go into DESIGN.md."""
import sys, time, json, numpy as np
sys.path.insert(0, "visual-odometry-project_amd")
from vo import _native, synthetic
H, W, N, hyp, F = 2160, 3840, 8000, 4000, 4
ctx = _native.Context(0)
stream = synthetic.Stream(F, H, W)
pipe = _native.Pipeline(ctx, H, W, F, stream.K, n_keypoints=N, klt_win=15, klt_max_level=3, hyp=hyp,
                        p3p_threshold=1.0, max_iterations=4000, refine_iters=20)
for i in range(F):
    pipe.set_frame(i, stream.image(i), stream.depth(i), stream.T_world_cam(i))
order = stream.order(400)
pipe.prime(order[0])
pos = 0
def run(n):
    global pos
    pipe.submit(order[pos], order[pos + 1])
    out = []
    for k in range(n):
        pos += 1
        if k + 1 < n:
            pipe.submit(order[pos], order[pos + 1])
        out.append(pipe.collect())
    return out
run(20)
ctx.prof_enable(-1); pipe.prof_reset()
run(8)
per = {}
for kid in range(_native.K_COUNT):
    ms, n = pipe.prof_read(kid)
    if n: per[ctx.kernel_name(kid)] = round(ms / n * 1e3, 1)
ctx.prof_disable()
ctx.sync()
t0 = time.perf_counter()
res = run(200)
ctx.sync()
dt = time.perf_counter() - t0
px = H * W
print(json.dumps({"frames_per_s": round(200 / dt, 1), "ms_per_step": round(dt / 200 * 1e3, 3), "per_kernel_us": per,
                  "tracked_median": float(np.median([r.n_tracked for r in res])),
                  "inliers_median": float(np.median([r.n_inliers for r in res])),
                  "harris_response_GBps": round(px * 9 / per["harris_response"] / 1e3, 1),
                  "nms_candidates_GBps": round(px * 8 / per["nms_candidates"] / 1e3, 1)}))
