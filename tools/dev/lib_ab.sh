#!/bin/bash
# same-box A/B of the built library against tools/dev/libvo_hip_p.so (a variant build, not kept): headline chain medians
set -eo pipefail
export VO_SYNTH_CACHE=/tmp/vo_synth_cache
L=visual-odometry-project_amd/vo/lib/libvo_hip.so
cp $L /tmp/lib_a.so
for round in 1 2; do
  for v in a p; do
    if [ $v = a ]; then cp /tmp/lib_a.so $L; else cp tools/dev/libvo_hip_p.so $L; fi
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api --no-legs --steps 1500 --warmup 100 > gpurun_out/ab_$v.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); c=d['chain_us']; print('$v', d['value'], {k: round(c[k],1) for k in ('tracker_start_to_regroup_start','regroup_to_next_tracker_start','step_period','regroup_to_hypotheses','hypotheses_to_pose','pose_to_landmarks','landmarks_to_record')})"
  done
done
cp /tmp/lib_a.so $L
