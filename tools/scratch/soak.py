"""Scratch: many steps of the cfg-2 pipeline with one frame of look-ahead; every result is checked against the
stream's analytic pose (catches rare paths: flagged draws -> sequential sampler, look-ahead ring restarts)."""
import sys, time, numpy as np
sys.path.insert(0, "visual-odometry-project_amd")
from vo import _native, synthetic
H, W, N, hyp, F = 1241, 1376, 2000, 1000, 8
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ctx = _native.Context(0)
stream = synthetic.Stream(F, H, W)
pipe = _native.Pipeline(ctx, H, W, F, stream.K, n_keypoints=N, klt_win=15, klt_max_level=2, hyp=hyp,
                        p3p_threshold=1.0, max_iterations=1000, refine_iters=20)
for i in range(F):
    pipe.set_frame(i, stream.image(i), stream.depth(i), stream.T_world_cam(i))
order = stream.order(steps + 8)
Tcw = [np.linalg.inv(stream.T_world_cam(i)) for i in range(F)]
pipe.prime(order[0])
bad = 0
worst_r = worst_t = 0.0
iters = []
t0 = time.perf_counter()
pipe.submit(order[0], order[1])
for k in range(steps):
    if k + 1 < steps:
        pipe.submit(order[k + 1], order[k + 2])
    r = pipe.collect()
    b = order[k + 1]
    R, t = np.array(r.R_refined).reshape(3, 3), np.array(r.t_refined)
    er, et = np.abs(R - Tcw[b][:3, :3]).max(), np.abs(t - Tcw[b][:3, 3]).max()
    worst_r, worst_t = max(worst_r, er), max(worst_t, et)
    iters.append(r.ransac_iterations)
    if r.best_index < 0 or r.n_tracked < 1500 or r.n_inliers < 300 or er > 5e-3 or et > 0.1:
        bad += 1
        if bad < 5:
            print("bad step", k, r.n_tracked, r.n_inliers, r.best_index, er, et)
dt = time.perf_counter() - t0
print("steps", steps, "bad", bad, "worst rot %.2e trans %.3f" % (worst_r, worst_t), "iters median", np.median(iters),
      "max", max(iters), "%.0f steps/s incl. python checks" % (steps / dt))
pipe.close()
