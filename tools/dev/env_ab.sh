#!/bin/bash
# same-box A/B of environment knobs: tools/dev/env_ab.sh "A=1,B=2" "A=0" ...   (each argument one variant: comma-separated
# assignments, "-" for none; headline chain medians, two rounds)
set -eo pipefail
export VO_SYNTH_CACHE=/tmp/vo_synth_cache
for round in 1 2; do
  for v in "$@"; do
    assigns=$(echo "$v" | tr ',' ' '); [ "$v" = "-" ] && assigns=""
    tag=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
    env $assigns timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api --no-legs --steps ${AB_STEPS:-1500} --warmup 100 ${AB_ARGS} > gpurun_out/ab_$tag.json 2>gpurun_out/ab_$tag.err
    python3 -c "
import json; d=json.load(open('gpurun_out/ab_$tag.json')); c=d['chain_us']; print('%-28s' % '$v', d['value'], {k[:14]: round(c[k],1) for k in ('tracker_start_to_regroup_start','regroup_to_next_tracker_start','step_period','regroup_to_hypotheses','hypotheses_to_pose','pose_to_landmarks','landmarks_to_record','record_to_next_regroup')}, d['loop']['steps_finished_by_host_path'])"
  done
done
