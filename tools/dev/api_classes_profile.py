"""cProfile of vo.driver.run (the class-per-stage drop-in loop) at the configuration's frame size."""
import cProfile, os, pstats, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "visual-odometry-project_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("OMP_NUM_THREADS", "4"); os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")
import numpy as np
from vo import driver
from vo.primitives import Sequence
seq = Sequence("synthetic", n_frames=30, height=1241, width=1376, channels=1)
pr = cProfile.Profile()
pr.enable()
r = driver.run(seq, "klt")
pr.disable()
print("frames/s", 1.0 / np.median(r["frame_seconds"]))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
