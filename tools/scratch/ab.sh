run() { timeout -k 10 200 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['per_kernel_us']['p3p_solve'])"; }
for rep in 1 2 3; do
cp tools/scratch/old/libvo_new.so visual-odometry-project_amd/vo/lib/libvo_hip.so; run new
cp tools/scratch/old/libvo_hip.so visual-odometry-project_amd/vo/lib/libvo_hip.so; run old
done
