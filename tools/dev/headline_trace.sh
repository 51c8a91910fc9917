#!/bin/bash
# a slice of the headline run's kernel trace: which kernel of which queue runs when (development measurement)
set -eo pipefail
out=gpurun_out/headline_trace
mkdir -p $out
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache VO_BENCH_RENDER_WORKERS=0
timeout -k 10 400 python3 tools/prerender_frames.py > $out/prerender.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-api --no-legs > $out/log.txt 2>&1
python3 - $out/t $out/slice.txt <<'PY'
import sys, glob, csv
fn = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
mid = len(rows) // 2
while "state_regroup_klt" not in rows[mid]["Kernel_Name"]: mid += 1
t0 = int(rows[mid]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    for r in rows[mid - 3:mid + 45]:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:34]
        o.write("%8.1f -> %8.1f (%6.1f us)  q%-3s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), n))
PY
rm -rf $out/t
cat $out/slice.txt
