#!/bin/bash
# Regenerates the material under profiles/ on a GPU box (run from the repo root):
#   tools/run_profiles.sh r03
# 1. kernel-trace statistics of the bench command (headline mode: one sequence, forward stream, detector gated);
# 2./3. one PMC pass each for FETCH_SIZE and WRITE_SIZE (never combined with a trace), with the detector on every
#    frame so that every kernel's figures are per executed launch; 4. kernel-trace statistics of that mode and of the
#    16-sequence run; 4b. the same runs with every pipeline stream mapped onto the main stream (each kernel's duration
#    without the others beside it); 5. the bench lines: headline (with the CPU baseline, API and in-line legs), 16
#    sequences per GPU, detector on every frame, cfg-3, cfg-5.
set -eo pipefail
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
export VO_BENCH_PROFILE_TAG=$tag
# the synthetic frames once, by worker processes outside the profiler (every run below reads them from the cache)
export VO_SYNTH_CACHE=/tmp/vo_synth_cache
timeout -k 10 400 python3 tools/prerender_frames.py >> gpurun_out/prof_progress.log 2>&1
export VO_BENCH_RENDER_WORKERS=0
echo "frames cached" >> gpurun_out/prof_progress.log
cmd="python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-api --no-legs"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $cmd > $out/stats.log 2>&1; echo stats >> gpurun_out/prof_progress.log
export VO_BENCH_DETECT_MARGIN=-1
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $cmd > $out/fetch.log 2>&1; echo fetch >> gpurun_out/prof_progress.log
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $cmd > $out/write.log 2>&1; echo write >> gpurun_out/prof_progress.log
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_all -- $cmd > $out/stats_all.log 2>&1; echo stats_all >> gpurun_out/prof_progress.log
unset VO_BENCH_DETECT_MARGIN
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_s16 -- $cmd --sequences 16 --steps 150 > $out/stats_s16.log 2>&1; echo stats_s16 >> gpurun_out/prof_progress.log
VO_ONE_STREAM=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_one -- $cmd > $out/stats_one.log 2>&1; echo stats_one >> gpurun_out/prof_progress.log
VO_ONE_STREAM=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_s16_one -- $cmd --sequences 16 --steps 100 > $out/stats_s16_one.log 2>&1; echo stats_s16_one >> gpurun_out/prof_progress.log
VO_ONE_STREAM=1 VO_BENCH_DETECT_MARGIN=-1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_s16_one_all -- $cmd --sequences 16 --steps 100 > $out/stats_s16_one_all.log 2>&1; echo stats_s16_one_all >> gpurun_out/prof_progress.log
VO_BENCH_CONFIG=cfg3 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_cfg3 -- $cmd --steps 60 --warmup 10 > $out/stats_cfg3.log 2>&1; echo stats_cfg3 >> gpurun_out/prof_progress.log
cp "$(ls $out/stats_cfg3/*/*kernel_stats.csv | head -1)" $out/${tag}_cfg3_kernel_stats.csv
cp "$(ls $out/stats_one/*/*kernel_stats.csv | head -1)" $out/${tag}_one_stream_kernel_stats.csv
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
timeout -k 10 300 python3 bench.py > $out/${tag}_bench_line.json 2> $out/bench.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/${tag}_bench_line_driver_flags.json 2> $out/bench_drv.err
timeout -k 10 300 python3 bench.py --sequences 16 --steps 300 --warmup 30 --no-cpu-baseline --no-api --no-legs > $out/${tag}_s16_bench_line.json 2> $out/bench_s16.err
VO_BENCH_DETECT_MARGIN=-1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-api --no-legs > $out/${tag}_detect_every_frame_bench_line.json 2> $out/bench_da.err
VO_BENCH_CONFIG=cfg3 timeout -k 10 300 python3 bench.py > $out/${tag}_cfg3_bench_line.json 2> $out/bench_cfg3.err
VO_BENCH_CONFIG=harris timeout -k 10 300 python3 bench.py > $out/${tag}_harris_bench_line.json 2> $out/bench_harris.err
VO_BENCH_CONFIG=cfg5 timeout -k 10 300 python3 bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-api --no-legs > $out/${tag}_cfg5_bench_line.json 2> $out/bench_cfg5.err
for f in $out/${tag}*_bench_line*.json; do echo "$f: $(grep -o '"value": [0-9.]*' $f | head -1)"; done
# only the condensed files travel back (the raw traces are tens of MB)
mkdir -p gpurun_out/profiles_$tag
cp $out/${tag}* gpurun_out/profiles_$tag/
rm -rf "$out"
