"""Prints the figures of bench lines that matter while tuning: python tools/dev/bsum.py gpurun_out/b_s1.json ..."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e)
        continue
    print("==", f, d["value"], "fps", d["ms_per_step"], "ms/step; dominant", d["roofline"]["kernel"],
          d["roofline"]["avg_launch_us"], "us frac", d["roofline"].get("frac"))
    print("  kernels", d["per_kernel_us"])
    print("  streaming", {k: (v["avg_launch_us"], v["frac"]) for k, v in d["roofline_streaming"].items()})
    c = d["chain_us"]
    print("  chain", {k: round(c[k], 1) for k in c if isinstance(c[k], float)}, c.get("pose_kernel_replay_refine_candidates"))
    print("  loop", d["loop"])
