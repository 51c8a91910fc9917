run() { timeout -k 10 200 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'])"; }
for rep in 1 2; do
  run base
  VO_PYR_WAIT=1 run host_wait
done
VO_DEBUG_TIMING=1 timeout -k 10 100 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>&1 | grep vo_pipeline
VO_PYR_WAIT=1 VO_DEBUG_TIMING=1 timeout -k 10 100 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>&1 | grep vo_pipeline
