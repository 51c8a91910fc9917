"""Where does the tracker's time go?  Stand-alone vo_klt_track at the configuration's shape with the iteration limit
and the number of levels varied (HIP-event time of the kernel alone)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "visual-odometry-project_amd")):
    sys.path.insert(0, p)
import numpy as np
from vo import _native, synthetic
ctx = _native.Context(0)
st = synthetic.Stream(2, 1241, 1376)
a, b = st.image(0), st.image(1)
kp = ctx.harris_keypoints(a, 9, 0.09, 2000, 5).astype(np.float32)
pts = np.concatenate([kp, kp[:450] + 0.5])
KID = [k for k in range(_native.K_COUNT) if ctx.kernel_name(k) == "klt_track"][0]
for lvl in (0, 1, 2):
    for it in (1, 2, 4, 10):
        ctx.klt_track(a, b, pts, win=15, max_level=lvl, max_iter=it)
        ctx.prof_enable(KID); ctx.prof_reset()
        for _ in range(10):
            o, s, e = ctx.klt_track(a, b, pts, win=15, max_level=lvl, max_iter=it)
        ms, n = ctx.prof_read(KID); ctx.prof_disable()
        print("levels %d max_iter %2d: %.1f us  (tracked %d)" % (lvl + 1, it, ms / n * 1e3, int(s.sum())), flush=True)
