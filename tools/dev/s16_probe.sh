#!/bin/bash
# 16-sequence run: kernel statistics and the bench line (development measurement)
set -eo pipefail
out=gpurun_out/s16_probe
mkdir -p $out
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache
timeout -k 10 400 python3 tools/prerender_frames.py > $out/prerender.log 2>&1
export VO_BENCH_RENDER_WORKERS=0
cmd="python3 bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-api --no-legs --sequences 16"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $cmd > $out/stats.log 2>&1
cp "$(ls $out/stats/*/*kernel_stats.csv | head -1)" $out/kernel_stats.csv
rm -rf $out/stats
timeout -k 10 300 $cmd --steps 300 > $out/bench_s16.json 2> $out/bench_s16.err
grep -o '"value": [0-9.]*' $out/bench_s16.json
head -12 $out/kernel_stats.csv
