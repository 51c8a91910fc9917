"""One frame through vo_sift a few times (development measurement: run under rocprofv3)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "visual-odometry-project_amd"))
from vo import _native, synthetic

if __name__ == "__main__":
    st = synthetic.Stream(100, 1241, 1376)
    img = np.ascontiguousarray(st.image(3))
    ctx = _native.default_context()
    for _ in range(6):
        kp, desc = ctx.sift(img, cap=2000)
    print(len(kp))
