#!/bin/bash
# bench with legs under stream settings
run() {
  timeout -k 10 400 python bench.py --steps 1000 --warmup 50 --no-api --no-cpu-baseline > gpurun_out/b8.json 2> gpurun_out/b8.err || tail -3 gpurun_out/b8.err
  python - <<PY
import json
d = json.load(open("gpurun_out/b8.json"))
c = d["chain_us"]
print("$1:", d["value"], "period med %.1f mean %.1f" % (c["step_period"], c["step_period_mean"]), "| every-frame", d["detector_every_frame"]["frames_per_s"], "| S16", d["sequences_16"]["frames_per_s"], "| S16 every-frame", d["sequences_16_detector_every_frame"]["frames_per_s"])
PY
}
run "no mask"
VO_SIDE_CUS=32-255 run "side 32-255"
VO_SIDE_CUS=16-255 run "side 16-255"
VO_SIDE_CUS=64-255 run "side 64-255"
