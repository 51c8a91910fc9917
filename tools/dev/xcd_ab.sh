#!/bin/bash
# needs tools/dev/libvo_hip_noxcd.so: libvo_hip.so built from a copy of csrc/ in which vo_xcd_tile returns its id
# (harris.o and klt.o rebuilt, the other objects reused; the variant library is not kept)
# A/B of the XCD-aware tile order (development measurement): the built library against tools/dev/libvo_hip_noxcd.so
set -eo pipefail
out=gpurun_out/xcd_ab
mkdir -p $out
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache
timeout -k 10 400 python3 tools/prerender_frames.py > $out/prerender.log 2>&1
export VO_BENCH_RENDER_WORKERS=0 VO_BENCH_DETECT_MARGIN=-1
cmd="python3 bench.py --warmup 20 --no-cpu-baseline --no-api --no-legs"
for v in xcd noxcd; do
  if [ $v = noxcd ]; then cp tools/dev/libvo_hip_noxcd.so visual-odometry-project_amd/vo/lib/libvo_hip.so; fi
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$v -- $cmd --steps 100 > $out/fetch_$v.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s1_$v -- $cmd --steps 200 > $out/s1_$v.log 2>&1
  VO_ONE_STREAM=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s16_$v -- $cmd --steps 60 --sequences 16 > $out/s16_$v.log 2>&1
  timeout -k 10 200 $cmd --steps 300 > $out/bench_s1_$v.json 2> $out/bench_s1_$v.err
  timeout -k 10 200 $cmd --steps 150 --sequences 16 > $out/bench_s16_$v.json 2> $out/bench_s16_$v.err
  python3 - $out $v <<'PY'
import sys, glob, csv, collections
sys.path.insert(0, "tools")
from summarize_profiles import pmc, short
out, v = sys.argv[1:3]
f = pmc(f"{out}/fetch_{v}", "FETCH_SIZE")
with open(f"{out}/summary_{v}.txt", "w") as o:
    for k in sorted(f, key=lambda k: -f[k][0])[:12]:
        o.write("fetch %-36s %6d launches %10.1f KiB raw (x2 = %.2f MB)\n" % (k, f[k][1], f[k][0], 2 * f[k][0] * 1024 / 1e6))
    for leg in ("s1", "s16"):
        for fn in glob.glob(f"{out}/{leg}_{v}/**/*kernel_stats.csv", recursive=True):
            rows = sorted(csv.DictReader(open(fn)), key=lambda r: -float(r["Percentage"]))[:14]
            for r in rows:
                o.write("%s %-36s %5s calls avg %8.1f us\n" % (leg, short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $out/fetch_$v $out/s1_$v $out/s16_$v
  echo "$v: s1 $(grep -o '"value": [0-9.]*' $out/bench_s1_$v.json) s16 $(grep -o '"value": [0-9.]*' $out/bench_s16_$v.json)"
done
