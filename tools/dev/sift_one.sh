#!/bin/bash
# one frame through vo_sift under the kernel trace: durations of the per-frame kernels (development measurement)
set -eo pipefail
out=gpurun_out/sift_one
mkdir -p $out
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 tools/dev/sift_one.py > $out/log.txt 2>&1
python3 - $out/t <<'PY'
import sys, glob, csv
fn = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "upsample2" in r["Kernel_Name"])
for r in rows[last:]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 12: print("%8.1f us grid %sx%s %s" % (d, r["Grid_Size_X"], r["Grid_Size_Y"], n))
print("frame: %.1f us" % ((int(rows[-1]["End_Timestamp"]) - int(rows[last]["Start_Timestamp"])) / 1e3))
PY
rm -rf $out/t
