"""Dev: per-launch durations of wgrad_kernel in the last backward of a rocprofv3 kernel trace of dev_wgrad_time.py.
usage: dev_wgrad_layers.py <kernel_trace.csv>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'wgrad_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = len(rows) // 6                      # forward-less script: 1 + 2 + 3 backward passes
last = rows[-per:]
tot = 0.0
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
    tot += d
    print('%-40s grid %-6s x %-5s %.3f ms' % (r['Kernel_Name'].split('(')[0][-40:], r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', '?'), d))
print('total %.3f ms' % tot)
