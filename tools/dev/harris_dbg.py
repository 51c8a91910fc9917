import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "visual-odometry-project_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from vo import _native, synthetic
from pipeline_oracle import OracleLoop, initial_harris_features
from oracle import harris_np, native
ctx = _native.Context(0)
H, W, N, F = 480, 640, 500, 3
stream = synthetic.Stream(F, H, W)
feats, T = initial_harris_features(stream, 0, N)
pipe = _native.Pipeline(ctx, H, W, F, stream.K, n_keypoints=N, hyp=256, p3p_threshold=1.0, max_iterations=1000, refine_iters=20, tracker="harris")
for i in range(F):
    pipe.set_frame(i, stream.image(i))
pipe.set_state(0, feats, T, T)
orc = OracleLoop(stream, N, 15, 2, refine_iters=20, tracker="harris")
orc.set_state(0, feats, T, T)
ref = orc.step(1)
r = pipe.step(0, 1)
st = pipe.get_state()
f = ref["features"]
print("device n_tri", r.n_triangulated, "oracle", ref["n_tri"], "pairs oracle", orc.n_pairs, "n2", st["n"], f.length)
print("kp equal", np.array_equal(st["keypoints"], f.keypoints.astype(np.float64)), "state equal", np.array_equal(st["state"], f.state))
# matched counts by state
print("device states", np.bincount(st["state"].astype(int), minlength=3), "oracle", np.bincount(f.state.astype(int), minlength=3))
# direct match through the host API for the same descriptor sets
img1 = stream.image(1)
kp2 = harris_np.nms_keypoints_fast(harris_np.harris_scores(img1, 9, 0.09), N, 5)
d2 = harris_np.patch_descriptors(img1, kp2, 9).reshape(N, -1).astype(np.float32)
d1 = feats.descriptors.reshape(N, -1).astype(np.float32)
pg = ctx.match_knn2_ratio(d1, d2, 0.85)
po = native.match_knn2_ratio(d1, d2, 0.85)[0]
print("host API pairs", len(pg), "oracle pairs", len(po), np.array_equal(pg, po))
kpd = pipe.get_detection() if False else None
