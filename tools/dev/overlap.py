import csv, glob, sys, collections
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0][:28], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[0][0]
# last 40 steps worth of rows
tail = rows[-600:]
q = collections.Counter((r[2], r[3]) for r in tail)
for k, v in sorted(q.items()):
    print(k, v)
print("---- timeline (us) of the last ~3 steps")
for s, e, n, qid, sid in rows[-45:]:
    print("%10.1f %8.1f  q=%s s=%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qid, sid, n))
