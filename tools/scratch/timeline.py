import csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/prof3/**/*kernel_trace.csv', recursive=True), key=lambda p: __import__('os').path.getmtime(p))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# pick a step in the middle: find klt kernels
klt = [i for i,n in enumerate(names) if 'klt_track' in n]
i0 = klt[len(klt)//2]; i1 = klt[len(klt)//2+2]
t0 = int(rows[i0]['Start_Timestamp'])
print("step duration (klt to klt): %.1f us" % ((int(rows[klt[len(klt)//2+1]]['Start_Timestamp'])-t0)/1e3))
for r in rows[i0-6:i1+1]:
    nm = r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0].replace('void ','')[:28]
    print("q%-3s %-28s start %8.1f  end %8.1f  dur %6.1f" % (r.get('Queue_Id','?'), nm, (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
