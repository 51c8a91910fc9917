"""Scratch: per-phase cycle counts inside klt_track_kernel (needs a build with -DVO_KLT_STAMPS)."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, "visual-odometry-project_amd")
from vo import _native, synthetic
H, W, N = 1241, 1376, 2000
ctx = _native.Context(0)
st = synthetic.Stream(3, H, W)
kp = ctx.harris_keypoints(st.image(0), 9, 0.09, N, 5)[0] if hasattr(ctx, "harris_keypoints") else None
from vo.features.harris import HarrisCornerDetector
import vo
lib = _native.load()
a, b = st.image(0), st.image(1)
kp = np.zeros((N, 2)); sc = None
rc = lib.vo_harris_keypoints(ctx._h, a.ctypes.data_as(C.c_void_p), H, W, 9, C.c_double(0.09), N, 5, kp.ctypes.data_as(C.c_void_p), None)
assert rc == 0
pts = kp.astype(np.float32)
out = np.zeros((N, 2), np.float32); stt = np.zeros(N, np.uint8); err = np.zeros(N, np.float32)
for rep in range(3):
    rc = lib.vo_klt_track(ctx._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), H, W, pts.ctypes.data_as(C.c_void_p), N, 15, 2, 10,
                          C.c_double(0.03), C.c_double(1e-4), out.ctypes.data_as(C.c_void_p), stt.ctypes.data_as(C.c_void_p), err.ctypes.data_as(C.c_void_p))
    assert rc == 0
s = np.zeros((N, 32), np.uint64)
lib.vo_debug_klt_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.vo_debug_klt_stamps(s.ctypes.data_as(C.c_void_p), N) == 0
s = s.astype(np.int64)
t0 = s[:, 31]
print("tracked", stt.sum(), "total cycles median", np.median(s[:, 30] - t0))
for L in (2, 1, 0):
    base = L * 8
    d = lambda x, y: np.median(s[:, base + x] - s[:, base + y])
    print("level %d: stageI %6.0f  template %6.0f  first-iter(+stageJ) %6.0f  rest-iters %6.0f  iters median %.1f mean %.2f  level total %6.0f" % (
        L, d(1, 0), d(2, 1), d(3, 2), d(4, 3), np.median(s[:, base + 5]), s[:, base + 5].mean(), d(4, 0)))
print("tail (error) %6.0f" % np.median(s[:, 30] - s[:, 4]))
