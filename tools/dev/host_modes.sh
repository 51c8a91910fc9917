#!/bin/bash
# bench with legs under detector-prediction settings
run() {
  timeout -k 10 400 python bench.py --steps 1000 --warmup 50 --no-api --no-cpu-baseline > gpurun_out/b8.json 2> gpurun_out/b8.err || tail -3 gpurun_out/b8.err
  python - <<PY
import json
d = json.load(open("gpurun_out/b8.json"))
c = d["chain_us"]
l = d["loop"]; s = d["sequences_16"]
print("$1:", d["value"], "det %.2f host %d %s" % (l["detector_executed_fraction_of_steps"], l["steps_finished_by_host_path"], l["host_path_reasons"]), "| S16", s["frames_per_s"], "det %.2f host %d %s" % (s["detector_executed_fraction_of_steps"], s["steps_finished_by_host_path"], s["host_path_reasons"]))
PY
}
VO_DETECT_LOSSES=4 VO_BENCH_DETECT_MARGIN=0.02 run "4 losses, margin .02"
VO_DETECT_LOSSES=2.5 VO_BENCH_DETECT_MARGIN=0.01 run "2.5 losses, margin .01"
VO_DETECT_LOSSES=2 VO_BENCH_DETECT_MARGIN=0.005 run "2 losses, margin .005"
VO_DETECT_LOSSES=3 VO_BENCH_DETECT_MARGIN=0.01 run "3 losses, margin .01"
