"""Scratch: BASELINE.json configs[2] stages (SIFT detect+describe, brute-force L2 2-NN + ratio) on a cfg-2
frame pair through the host ABI (copies included); event-timed kernel groups.  Not the bench line."""
import sys, time, numpy as np
sys.path.insert(0, "visual-odometry-project_amd")
from vo import _native, synthetic
H, W = 1241, 1376
ctx = _native.Context(0)
st = synthetic.Stream(2, H, W)
a, b = st.image(0), st.image(1)
for _ in range(2):
    ka, da = ctx.sift(a); kb, db = ctx.sift(b)
ctx.prof_enable(-1); ctx.prof_reset()
t0 = time.perf_counter(); ka, da = ctx.sift(a); t1 = time.perf_counter()
# strongest 2000 by response, as cfg-3 caps the keypoints
ia, ib = np.argsort(-ka[:, 4])[:2000], np.argsort(-kb[:, 4])[:2000]
qa, qb = da[ia], db[ib]
sift_prof = {ctx.kernel_name(k): ctx.prof_read(k) for k in range(_native.K_COUNT)}
ctx.match_knn2_ratio(qa, qb, 0.8)          # warm (code object, attribute)
ctx.prof_reset()
t2 = time.perf_counter()
for _ in range(10):
    m = ctx.match_knn2_ratio(qa, qb, 0.8)
t3 = time.perf_counter()
print("sift: %d keypoints, %.2f ms per frame (host call, copies included); match 2000x2000x128: %d pairs, %.3f ms per host call" %
      (len(ka), (t1 - t0) * 1e3, len(m), (t3 - t2) * 1e2))
for name, (ms, n) in sift_prof.items():
    if n and name.startswith("sift"): print("  ", name, "%.1f us total over %d launches" % (ms * 1e3, n))
for kid in range(_native.K_COUNT):
    ms, n = ctx.prof_read(kid)
    if n: print("  ", ctx.kernel_name(kid), "%.1f us per launch (%d warm launches)" % (ms * 1e3 / n, n))
