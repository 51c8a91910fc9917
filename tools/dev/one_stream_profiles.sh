#!/bin/bash
# kernel durations without overlap (VO_ONE_STREAM=1) at 1 and 16 sequences, and 16 with the detector on every frame
set -eo pipefail
export TMPDIR=/tmp
o=gpurun_out/p1
mkdir -p $o
cmd="python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-api --no-legs"
VO_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $o/s1 -- $cmd > $o/s1.log 2>&1
VO_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $o/s16 -- $cmd --sequences 16 --steps 100 > $o/s16.log 2>&1
VO_ONE_STREAM=1 VO_BENCH_DETECT_MARGIN=-1 rocprofv3 --kernel-trace --stats --output-format csv -d $o/s16a -- $cmd --sequences 16 --steps 100 > $o/s16a.log 2>&1
for d in s1 s16 s16a; do cp "$(ls $o/$d/*/*kernel_stats.csv | head -1)" $o/$d.csv; done
