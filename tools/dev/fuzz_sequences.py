"""python tools/dev/fuzz_sequences.py [seed [trials [big]]]: the randomised several-sequences check of tests/pipeline_fuzz.py."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "visual-odometry-project_amd"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
from vo import _native
from pipeline_fuzz import run_trials
ctx = _native.Context(0)
print("failures:", run_trials(ctx, int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 12, True, len(sys.argv) > 3))
