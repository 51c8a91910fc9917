#!/bin/bash
# the 15x15 tracker with 32 lanes per keypoint against 16: parity tests under the knob, then headline and 16-sequence A/B
set -eo pipefail
export VO_SYNTH_CACHE=/tmp/vo_synth_cache
VO_KLT_LPK=32 timeout -k 10 300 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_pipeline.py -x -q -m gpu -k "klt or oracle_loop or sequences" 2>&1 | tail -3
tools/dev/env_ab.sh VO_KLT_LPK 16 32
for round in 1 2; do
  for v in 16 32; do
    VO_KLT_LPK=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api --no-legs --sequences 16 --steps 200 --warmup 30 > gpurun_out/ab_lpk16_$v.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('gpurun_out/ab_lpk16_$v.json')); print('S16 LPK=$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])"
  done
done
