import sys; sys.path.insert(0, "visual-odometry-project_amd")
import numpy as np
from vo import driver
from vo.primitives import Sequence
for mode, ch in (("klt", 3), ("harris", 1)):
    seq = Sequence("synthetic", n_frames=20, height=480, width=640, channels=ch)
    out = driver.run(seq, mode, verbose=False)
    err = driver.trajectory_error(out, seq)
    traj = out["trajectory"]
    print(mode, err, "ms/frame %.1f" % (1e3*out["frame_seconds"].mean()), "landmarks", out["n_landmarks"][:8])
    frames = [0] + list(range(2, 2 + len(traj) - 1))
    for i in (1, 2, 5, 10, len(traj)-1):
        gt = np.linalg.inv(seq.ground_truth_pose(0)) @ seq.ground_truth_pose(frames[i])
        print("  frame", frames[i], "est*s", np.round(traj[i][:3,3]*err["scale"],3), "gt", np.round(gt[:3,3],3))
