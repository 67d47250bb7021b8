"""Timeline of one image's kernels from a rocprofv3 kernel trace: python tools/trace_timeline.py <kernel_trace.csv> [n]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'mn_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mid = len(rows) // 2
while 'mn_cc_sign' not in rows[mid]['Kernel_Name']:
    mid += 1
t0 = int(rows[mid]['Start_Timestamp'])
last_end = {}
for r in rows[mid:mid + n]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r['Queue_Id']
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:26]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    print("q%s %-26s start %8.1f  dur %6.1f  gap on its queue %6.1f" % (q, name, (s - t0) / 1e3, (e - s) / 1e3, gap))
    last_end[q] = e
