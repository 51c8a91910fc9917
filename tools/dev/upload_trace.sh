#!/bin/bash
set -eo pipefail
out=gpurun_out/upload_trace
mkdir -p $out
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/t -- python3 tools/dev/upload_trace.py > $out/log.txt 2>&1
tail -3 $out/log.txt
python3 - $out/t <<'PY'
import sys, glob, csv
d = sys.argv[1]
ev = []
for fn in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r.get("Queue_Id", "?"), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40])))
for fn in glob.glob(d + "/*/*memory_copy_trace.csv"):
    rows = list(csv.DictReader(open(fn)))
    print("copy columns:", list(rows[0].keys()) if rows else None)
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s stream %s" % (r.get("Direction", "?"), r.get("Stream_Id", "?"))))
ev.sort()
# the middle of the first timed pass: the 60th large host-to-device copy
big = [i for i, e in enumerate(ev) if e[2].startswith("C") and "HOST_TO_DEVICE" in e[2] and e[1] - e[0] > 15000]
print("large copies:", len(big))
if big:
    i0 = big[min(len(big) - 1, 40)]
    t0 = ev[i0][0]
    with open(d + "/../slice.txt", "w") as o:
        for e in ev[i0 - 5:i0 + 70]:
            o.write("%9.1f %8.1f us  %s\n" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2]))
PY
rm -rf $out/t
cat $out/slice.txt
