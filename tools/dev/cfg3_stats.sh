#!/bin/bash
# cfg-3 (SIFT tracker mode): kernel statistics, multi-stream and one-stream (development measurement)
set -eo pipefail
out=gpurun_out/cfg3_stats
mkdir -p $out
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache
timeout -k 10 400 python3 tools/prerender_frames.py > $out/prerender.log 2>&1
export VO_BENCH_RENDER_WORKERS=0 VO_BENCH_CONFIG=cfg3
cmd="python3 bench.py --warmup 10 --steps 60 --no-cpu-baseline --no-api --no-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/a -- $cmd > $out/a.log 2>&1
cp "$(ls $out/a/*/*kernel_stats.csv | head -1)" $out/kernel_stats.csv
VO_ONE_STREAM=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/b -- $cmd > $out/b.log 2>&1
cp "$(ls $out/b/*/*kernel_stats.csv | head -1)" $out/kernel_stats_one_stream.csv
python3 - $out <<'PY'
import sys, glob, csv
out = sys.argv[1]
fn = glob.glob(out + "/b/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
mid = len(rows) // 2
with open(out + "/trace_slice_one_stream.txt", "w") as o:
    t0 = int(rows[mid]["Start_Timestamp"])
    for r in rows[mid:mid + 260]:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
        o.write("%9.1f %8.1f us  grid %-14s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]), n))
PY
rm -rf $out/a $out/b
