import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "visual-odometry-project_amd"))
import numpy as np
from vo import driver
from vo.primitives import Sequence
mode = sys.argv[1] if len(sys.argv) > 1 else "current"
seq = Sequence("synthetic", n_frames=24, height=480, width=640, channels=3)
out = driver.run_on_device(seq, n_keypoints=500, verbose=True, redetect_start_pose=mode)
traj = out["trajectory"]
for i, T in enumerate(traj):
    print(i, np.round(T[:3, 3], 3))
print(driver.trajectory_error(out, seq))
seq = Sequence("synthetic", n_frames=24, height=480, width=640, channels=3)
out = driver.run(seq, "klt", verbose=True)
for i, T in enumerate(out["trajectory"]):
    print(i, np.round(T[:3, 3], 3))
print(driver.trajectory_error(out, seq))
