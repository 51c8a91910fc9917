import csv, glob, sys
f = glob.glob('gpurun_out/prof2/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last NMS sequence
names = [r['Kernel_Name'] for r in rows]
last = max(i for i,n in enumerate(names) if 'nms_candidates' in n)
t0 = int(rows[last]['Start_Timestamp'])
for r in rows[last-1:last+16]:
    print("%-40s start %8.1f us  dur %8.2f us" % (r['Kernel_Name'].split('(')[0][-40:], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
