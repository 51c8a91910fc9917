#!/bin/bash
# kernel durations without overlap: all pipeline streams = the main stream (VO_ONE_STREAM=1), rocprofv3 kernel trace
set -eo pipefail
S=${1:-16}
out=gpurun_out/one_stream_s$S
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp VO_ONE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -- python3 bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-api --sequences $S > $out/log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/st/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:16]:
    n = r['Name']; n = n.split('::')[1][:36] if '::' in n else n[:36]
    print('%-38s calls %5s avg %9.1f min %8.1f max %8.1f us %5s%%' % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Percentage']))
PY
tail -1 $out/log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
