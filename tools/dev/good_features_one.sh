#!/bin/bash
set -eo pipefail
out=gpurun_out/gf_one
mkdir -p $out
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache
for mode in rounds walk; do
  if [ $mode = walk ]; then export VO_GREEDY_WALK=1; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$mode -- python3 tools/dev/good_features_one.py > $out/log_$mode.txt 2>&1
  echo "== $mode"; python3 - "$(ls $out/$mode/*/*kernel_stats.csv | head -1)" <<'PY'
import sys, csv
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-60s calls %3s avg %8.1f us" % (r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $out/$mode
done
