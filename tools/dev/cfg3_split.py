"""Where a cfg-3 frame's wall time goes through the host entry points: vo_sift, vo_match_knn2_ratio, Python around them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "visual-odometry-project_amd")]
import numpy as np
from vo import _native, synthetic


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "torch":
        import torch
        print("torch cuda", torch.cuda.is_available())
    ctx = _native.Context(0)
    st = synthetic.Stream(12, 1241, 1376).prefetch(workers=8)
    imgs = [st.image(i) for i in range(12)]
    kp, d0 = ctx.sift(imgs[0], cap=2000)
    for rep in range(2):
        ts = {"sift": 0.0, "match": 0.0}
        for i in range(1, 12):
            t = time.perf_counter()
            kp, d = ctx.sift(imgs[i], cap=2000)
            ts["sift"] += time.perf_counter() - t
            t = time.perf_counter()
            pairs = ctx.match_knn2_ratio(d0, d, 0.8)
            ts["match"] += time.perf_counter() - t
            d0 = d
        print({k: round(1e3 * v / 11, 3) for k, v in ts.items()}, len(d), len(pairs))
    import ctypes as C
    lib = ctx._lib
    img = np.ascontiguousarray(imgs[3])
    kpb, db, n = np.empty((2000, 6), np.float32), np.empty((2000, 128), np.float32), C.c_int32(0)
    t = time.perf_counter()
    for _ in range(20):
        lib.vo_sift(ctx._h, img.ctypes.data_as(C.c_void_p), 1241, 1376, 2000, kpb.ctypes.data_as(C.c_void_p), db.ctypes.data_as(C.c_void_p), C.byref(n))
    print("vo_sift alone ms", 1e3 * (time.perf_counter() - t) / 20)


if __name__ == "__main__":
    main()
