import sys, os, faulthandler
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "visual-odometry-project_amd"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import numpy as np
from vo import _native, synthetic
from pipeline_oracle import initial_features
ctx0 = _native.default_context()
H, W, N, hyp, F = 240, 320, 300, 256, 6
stream = synthetic.Stream(F, H, W)
feats, T = initial_features(stream, 0, N)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for rep in range(reps):
    ctx = _native.Context(0)
    pipe = _native.Pipeline(ctx, H, W, F, stream.K, n_keypoints=N, klt_win=15, klt_max_level=2, hyp=hyp, p3p_threshold=1.0,
                            max_iterations=1000, refine_iters=20)
    for i in range(F):
        pipe.set_frame(i, stream.image(i))
    pipe.set_state(0, feats, T, T)
    order = stream.order(7)
    for a, b in zip(order[:-1], order[1:]):
        r = pipe.step(a, b)
        st = pipe.get_state()
    pipe.close()
    ctx.close()
    print("rep", rep, r.n_tracked, r.n_inliers, flush=True)
