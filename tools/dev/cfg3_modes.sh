#!/bin/bash
run() {
  VO_BENCH_CONFIG=cfg3 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 150 > gpurun_out/b16.json 2> gpurun_out/b16.err || tail -3 gpurun_out/b16.err
  python - <<PY
import json
d = json.load(open("gpurun_out/b16.json"))
print("$1:", d["value"], d["ms_per_step"], d["per_kernel_us"].get("sift_scale_space"))
PY
  grep vo_pipeline gpurun_out/b16.err | tail -1
}
export VO_DEBUG_TIMING=1
run "default"
GPU_MAX_HW_QUEUES=8 run "8 hw queues"
VO_HOST_THREADS_BUDGET=1 run "budget 1"
GPU_MAX_HW_QUEUES=8 VO_SIFT_ONE_CONTEXT=1 run "8 hw queues, one context"
