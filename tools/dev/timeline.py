"""Per-step timeline from a rocprofv3 kernel trace of bench.py: for a few steps in the middle of the run, every
kernel's start / end relative to the step's regroup start.  Usage: timeline.py <kernel_trace.csv> [first_step] [count]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 150
count = int(sys.argv[3]) if len(sys.argv) > 3 else 3
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.match(r"[A-Za-z_0-9]+", n).group(0)
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows))
reg = [i for i, e in enumerate(ev) if e[2] == "state_regroup_klt_kernel"]
for k in range(first, first + count):
    i0, i1 = reg[k], reg[k + 1]
    t0 = ev[i0][0]
    print("---- step", k, "period %.1f us" % ((ev[i1][0] - t0) / 1e3))
    for s, e, n, q in ev[i0:i1 + 1]:
        print("  %-28s q%-3s %8.1f -> %8.1f  (%6.1f)" % (n, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
