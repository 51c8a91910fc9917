import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "visual-odometry-project_amd"))
import numpy as np
import bench
from vo import driver, synthetic
from vo.features.klt import KLTTracker
from vo.landmarks import LandmarksTriangulator
stream = synthetic.Stream(3, bench.H, bench.W)
for thr in (0.25, 0.5, 1.0, 2.0):
    for win in (15, 21):
        orig = driver.make_estimators
        def mk(camera, thr=thr):
            tri, pe = orig(camera)
            tri._ransac_reproj_threshold = thr
            return tri, pe
        driver.make_estimators = mk
        saved = (dict(KLTTracker._feature_params), dict(KLTTracker._lk_params))
        def setup():
            KLTTracker._feature_params = dict(saved[0], maxCorners=2000)
            KLTTracker._lk_params = dict(saved[1], winSize=(win, win), maxLevel=3 if win == 21 else 2)
        try:
            state, tracker, _, _ = driver.bootstrap(bench.ResidentSequence(stream), "klt", tracker_setup=setup)
        finally:
            KLTTracker._feature_params, KLTTracker._lk_params = saved
            driver.make_estimators = orig
        f = state.curr_frame.features
        T = state.curr_pose
        gt = np.linalg.inv(stream.T_world_cam(0)) @ stream.T_world_cam(2)
        t = T[:3, 3] / np.linalg.norm(T[:3, 3]) * 1.6
        print("thr", thr, "win", win, "features", f.length, "landmarks", int((f.state == 2).sum()), "rot err", np.linalg.norm(T[:3, :3] - gt[:3, :3]).round(5), "t err (m, scaled)", np.linalg.norm(t - gt[:3, 3]).round(4), flush=True)
