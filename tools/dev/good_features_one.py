"""vo_good_features at 1376x1241 a few times (development measurement: run under rocprofv3)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "visual-odometry-project_amd"))
from vo import _native, synthetic

if __name__ == "__main__":
    img = np.ascontiguousarray(synthetic.Stream(100, 1241, 1376).image(3))
    ctx = _native.default_context()
    for _ in range(5):
        pts = ctx.good_features(img, None, 2000, 0.01, 8, 7)
    print(len(pts))
