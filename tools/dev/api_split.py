"""Where does a frame of vo.driver.run_on_device go?  (collect / upload / submit, and how many steps the host path finished)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "visual-odometry-project_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("OMP_NUM_THREADS", "4"); os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")
import numpy as np
from vo import _native, _pipeline
from vo.primitives import Sequence
from vo import driver

T = {"collect": [], "set_frame": [], "submit": []}
for name in ("collect", "set_frame", "submit"):
    orig = getattr(_pipeline.Pipeline, name)
    def wrap(self, *a, _o=orig, _n=name, **k):
        t = time.perf_counter(); r = _o(self, *a, **k); T[_n].append(time.perf_counter() - t); return r
    setattr(_pipeline.Pipeline, name, wrap)
ctx = _native.Context(0)
seq = Sequence("synthetic", n_frames=40, height=1241, width=1376, channels=1)
r = driver.run_on_device(seq, n_keypoints=2000, klt_win=15, klt_max_level=2, hyp=int(os.environ.get('HYP', '1000')), context=ctx,
                         bootstrap_win=21, bootstrap_max_level=3, bootstrap_threshold=1.0)
import collections
print("recovery reasons", collections.Counter(x.reserved for x in r.get("results", [])))
print("frames/s", 1.0 / np.median(r["frame_seconds"]), "median ms", 1e3 * np.median(r["frame_seconds"]))
for k, v in T.items():
    print(k, "median us", 1e6 * np.median(v), "mean us", 1e6 * np.mean(v), "n", len(v))
