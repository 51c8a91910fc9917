run() { timeout -k 10 200 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline $2 2>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['roofline']['avg_launch_us'])" || tail -5 gpurun_out/ab.err; }
run plain
VO_BENCH_NO_AG=3 run exch3 --exchange
VO_BENCH_NO_AG=1 run exch1 --exchange
VO_BENCH_NO_AG=0 run exch0 --exchange
