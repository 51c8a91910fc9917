import sys, ctypes as C
sys.path.insert(0, "visual-odometry-project_amd"); sys.path.insert(0, ".")
import numpy as np
from vo import _native, synthetic
from oracle import harris_np
ctx = _native.Context(0)
i0, d0, T0, K = synthetic.render(3, 1241, 1376)
i1, d1, T1, K = synthetic.render(4, 1241, 1376)
kp = harris_np.nms_keypoints_fast(harris_np.harris_scores(i0), 2000, 5)[:, :, 0].astype(np.float32)
for _ in range(2): ctx.klt_track(i0, i1, kp, win=15, max_level=2)
d_st = ctx.alloc(2000 * 16 * 8)
lib = _native.load(); lib.vo_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.vo_debug_set_stamps(ctx._h, C.c_void_p(d_st))
out, st, err = ctx.klt_track(i0, i1, kp, win=15, max_level=2)
s = ctx.download(d_st, (2000, 16), np.uint64).astype(np.int64)
print("total cycles: mean %.0f median %.0f max %d" % (s[:,0].mean(), np.median(s[:,0]), s[:,0].max()))
for l in (2, 1, 0):
    n = np.maximum(s[:, 9+l], 1)
    print("level %d: template %.0f cycles; iterations mean %.2f; per-iteration %.0f cycles" % (l, s[:,1+l].mean(), s[:,9+l].mean(), (s[:,5+l]/n).mean()))
