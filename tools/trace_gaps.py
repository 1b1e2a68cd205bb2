"""Idle gaps of the GPU inside the LAST batched fit of a rocprofv3 --kernel-trace CSV (no kernel in flight for 0.5 ms or more):
when, how long, the kernel that ended before and the one that started after.  python tools/trace_gaps.py kernel_trace.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_scale_rows' in r['Kernel_Name']]
starts = [i for n, i in enumerate(idx) if n == 0 or int(rows[i]['Start_Timestamp']) - int(rows[idx[n - 1]]['Start_Timestamp']) > 200e6]
seg = rows[starts[-1]:]
t0 = int(seg[0]['Start_Timestamp'])


def nm(n):
    m = re.search(r'(k_\w+(<[^>]*>)?|Cijk_\w{0,20}|__amd\w+)', n)
    return m.group(1) if m else n[:40]


end = int(seg[0]['End_Timestamp'])
last = seg[0]
tot = 0.
for r in seg[1:]:
    s = int(r['Start_Timestamp'])
    if s > end:
        if s - end >= 0.5e6:
            print('%8.1f ms  idle %6.2f ms   after %-28s before %s (grid %s)' % ((end - t0) / 1e6, (s - end) / 1e6, nm(last['Kernel_Name']),
                                                                           nm(r['Kernel_Name']), r['Grid_Size_X']))
        tot += s - end
    if int(r['End_Timestamp']) > end:
        end = int(r['End_Timestamp'])
        last = r
print('idle in all: %.1f ms of %.1f' % (tot / 1e6, (end - t0) / 1e6))
