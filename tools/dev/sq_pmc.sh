#!/bin/bash
# SQ counters per kernel at 16 sequences, one stream, detector on every frame (development measurement)
set -eo pipefail
out=gpurun_out/sq_pmc
mkdir -p $out
export TMPDIR=/tmp VO_SYNTH_CACHE=/tmp/vo_synth_cache
timeout -k 10 400 python3 tools/prerender_frames.py > $out/prerender.log 2>&1
export VO_BENCH_RENDER_WORKERS=0 VO_BENCH_DETECT_MARGIN=-1 VO_ONE_STREAM=1
cmd="python3 bench.py --warmup 10 --steps 30 --no-cpu-baseline --no-api --no-legs --sequences ${1:-16}"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/a -- $cmd > $out/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d $out/b -- $cmd > $out/b.log 2>&1
python3 - $out <<'PY'
import sys, glob, csv, collections
sys.path.insert(0, "tools")
from summarize_profiles import short
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob(out + "/[ab]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = short(r["Kernel_Name"]); tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_WAVES"): n[(k, r["Counter_Name"])] += 1
with open(out + "/summary.txt", "w") as o:
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0))[:14]:
        t = tot[k]; wc = t.get("SQ_WAVE_CYCLES", 1) or 1; L = n[(k, "SQ_WAVE_CYCLES")] or 1
        o.write("%-34s launches %4d wave_cyc/launch %.3g | wait_any %.2f wait_inst %.2f active %.2f | valu %.2f lds %.2f wait_lds %.2f bankconf/ldsactive %.2f | vmem %.2f sca %.2f\n" % (
            k, L, wc / L, t["SQ_WAIT_ANY"] / wc, t["SQ_WAIT_INST_ANY"] / wc, t["SQ_ACTIVE_INST_ANY"] / wc, t["SQ_ACTIVE_INST_VALU"] / wc,
            t["SQ_ACTIVE_INST_LDS"] / wc, t["SQ_WAIT_INST_LDS"] / wc, t["SQ_LDS_BANK_CONFLICT"] / max(t["SQ_ACTIVE_INST_LDS"], 1),
            t["SQ_ACTIVE_INST_VMEM"] / wc, t["SQ_ACTIVE_INST_SCA"] / wc))
        L2 = n[(k, "SQ_WAVES")] or 1
        o.write("    per launch: waves %.0f insts valu %.3g salu %.3g lds %.3g vmem_rd %.3g vmem_wr %.3g\n" % (
            t["SQ_WAVES"] / L2, t["SQ_INSTS_VALU"] / L2, t["SQ_INSTS_SALU"] / L2, t["SQ_INSTS_LDS"] / L2, t["SQ_INSTS_VMEM_RD"] / L2, t["SQ_INSTS_VMEM_WR"] / L2))
PY
rm -rf $out/a $out/b
cat $out/summary.txt
