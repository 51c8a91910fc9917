import ctypes as C, sys, numpy as np
sys.path.insert(0, "visual-odometry-project_amd")
from vo import _native, synthetic
H, W, N, HYP = 1241, 1376, 2000, 1000
ctx = _native.Context(0)
st = synthetic.Stream(4, H, W)
pipe = _native.Pipeline(ctx, H, W, 4, st.K, n_keypoints=N, klt_win=15, klt_max_level=2, hyp=HYP, p3p_threshold=1.0, max_iterations=1000)
for i in range(4):
    pipe.set_frame(i, st.image(i), st.depth(i), st.T_world_cam(i))
o = st.order(6)
pipe.prime(o[0])
for a, b in zip(o[:-1], o[1:]):
    r = pipe.step(a, b)
lib = _native.load()
d = np.zeros((HYP, 8), np.uint64)
lib.vo_debug_p3p.argtypes = [C.c_void_p, C.c_int]
assert lib.vo_debug_p3p(d.ctypes.data_as(C.c_void_p), HYP) == 0
d = d.astype(np.int64)
seq = [(0, 6, "samples (host raws)"), (6, 1, "point loads"), (1, 7, "bearing vectors f"), (7, 2, "frame + coefficients"), (2, 3, "quartic"), (3, 4, "root -> pose")]
for a, b, nm in seq:
    dt = d[:, b] - d[:, a]
    ok = (d[:, b] > 0) & (d[:, a] > 0)
    print("%-22s median %7d  p99 %7d  max %7d ticks" % (nm, np.median(dt[ok]), np.percentile(dt[ok], 99), dt[ok].max()))
tot = d[:, 4] - d[:, 0]
print("total: median %d max %d" % (np.median(tot), tot.max()))
