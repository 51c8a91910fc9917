#!/bin/bash
# Regenerates the material under profiles/ on a GPU box (run from the repo root):
#   tools/run_profiles.sh r01
# 1. kernel-trace statistics of the bench command, 2./3. one PMC pass each for FETCH_SIZE and
# WRITE_SIZE (never combined with a trace), 4. the plain bench line (with the CPU baseline leg).
set -eo pipefail
tag=${1:-r02}
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cmd="python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-api"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $cmd > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $cmd > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $cmd > $out/write.log 2>&1
python3 tools/summarize_profiles.py $out/stats $out/fetch $out/write $out/$tag
cp "$(ls $out/stats/*/*kernel_stats.csv | head -1)" $out/${tag}_bench_kernel_stats.csv
# the bench line reads the PMC figures it reports as roofline.traffic from profiles/<tag>_pmc_traffic.json
cp $out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
python3 bench.py > $out/${tag}_bench_line.json 2> $out/bench.err
tail -c 2500 $out/${tag}_bench_line.json
