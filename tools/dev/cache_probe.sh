#!/bin/bash
# how long the frame cache takes to fill and to read on this box, and a rocprof run on top of it
export VO_SYNTH_CACHE=/tmp/vo_synth_cache
timeout -k 10 200 python3 tools/prerender_frames.py --small
export VO_BENCH_RENDER_WORKERS=0
export TMPDIR=/tmp
t0=$(date +%s)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/probe -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-api --no-legs > gpurun_out/probe.log 2>&1
echo "rocprof headline run: rc=$? $(( $(date +%s) - t0 )) s"
rm -rf gpurun_out/probe
