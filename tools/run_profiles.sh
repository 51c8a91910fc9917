#!/bin/bash
# Regenerates the material under profiles/ on a GPU box (run from the repo root):
#   tools/run_profiles.sh r02
# 1. kernel-trace statistics of the bench command (headline mode: one sequence, detector within the margin only);
# 2./3. one PMC pass each for FETCH_SIZE and WRITE_SIZE (never combined with a trace), with the detector on every
#    frame so that every kernel's figures are per executed launch; 4. kernel-trace statistics of that mode and of the
#    16-sequence run; 5. the bench lines: headline (with the CPU baseline and API legs), 4 and 16 sequences per GPU,
#    detector on every frame (1 and 16 sequences), cfg-3, cfg-5.  4b: see below.
set -eo pipefail
tag=${1:-r02}
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cmd="python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-api"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $cmd > $out/stats.log 2>&1
export VO_BENCH_DETECT_MARGIN=-1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $cmd > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $cmd > $out/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_all -- $cmd > $out/stats_all.log 2>&1
unset VO_BENCH_DETECT_MARGIN
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_s16 -- $cmd --sequences 16 --steps 100 > $out/stats_s16.log 2>&1
# 4b. the same 16-sequence run with every pipeline stream = the main stream (VO_ONE_STREAM=1): each kernel's duration
#     without the others running beside it (what a roofline fraction of a single kernel should be computed from)
VO_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_s16_one -- $cmd --sequences 16 --steps 60 --warmup 20 > $out/stats_s16_one.log 2>&1
VO_ONE_STREAM=1 VO_BENCH_DETECT_MARGIN=-1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_s16_one_all -- $cmd --sequences 16 --steps 60 --warmup 20 > $out/stats_s16_one_all.log 2>&1
cp "$(ls $out/stats_s16_one/*/*kernel_stats.csv | head -1)" $out/${tag}_s16_one_stream_kernel_stats.csv
cp "$(ls $out/stats_s16_one_all/*/*kernel_stats.csv | head -1)" $out/${tag}_s16_one_stream_detect_every_frame_kernel_stats.csv
python3 tools/summarize_profiles.py $out/stats $out/fetch $out/write $out/$tag
python3 tools/summarize_profiles.py $out/stats_all $out/fetch $out/write $out/${tag}_detect_every_frame
python3 tools/summarize_profiles.py $out/stats_s16 $out/fetch $out/write $out/${tag}_s16
rm -f $out/${tag}_detect_every_frame_pmc_traffic.json $out/${tag}_s16_pmc_traffic.json
cp "$(ls $out/stats/*/*kernel_stats.csv | head -1)" $out/${tag}_bench_kernel_stats.csv
cp "$(ls $out/stats_s16/*/*kernel_stats.csv | head -1)" $out/${tag}_s16_kernel_stats.csv
# the bench line reads the PMC figures it reports as roofline.traffic from profiles/<tag>_pmc_traffic.json
cp $out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
python3 bench.py > $out/${tag}_bench_line.json 2> $out/bench.err
python3 bench.py --sequences 4 --steps 500 --warmup 50 --no-cpu-baseline --no-api > $out/${tag}_s4_bench_line.json 2> $out/bench_s4.err
python3 bench.py --sequences 16 --steps 200 --warmup 30 --no-cpu-baseline --no-api > $out/${tag}_s16_bench_line.json 2> $out/bench_s16.err
VO_BENCH_DETECT_MARGIN=-1 python3 bench.py --no-cpu-baseline --no-api > $out/${tag}_detect_every_frame_bench_line.json 2> $out/bench_da.err
VO_BENCH_DETECT_MARGIN=-1 python3 bench.py --sequences 16 --steps 200 --warmup 30 --no-cpu-baseline --no-api > $out/${tag}_detect_every_frame_s16_bench_line.json 2> $out/bench_da16.err
VO_BENCH_CONFIG=cfg3 python3 bench.py > $out/${tag}_cfg3_bench_line.json 2> $out/bench_cfg3.err
VO_BENCH_CONFIG=cfg5 python3 bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-api > $out/${tag}_cfg5_bench_line.json 2> $out/bench_cfg5.err
for f in $out/${tag}*_bench_line.json; do echo "$f: $(grep -o '"value": [0-9.]*' $f | head -1)"; done
