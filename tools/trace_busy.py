"""GPU occupancy digest of a rocprofv3 --kernel-trace CSV over the LAST batched fit in it: wall span, time with at least one
kernel running, and the CU-time the K3 launches could have used (min(systems, 256) CUs x duration) against what their
systems needed.  Usage: python tools/trace_busy.py kernel_trace.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last fit starts at the last k_scale_rows burst: find the last gap > 50 ms before a k_scale_rows
idx = [i for i, r in enumerate(rows) if 'k_scale_rows' in r['Kernel_Name']]
starts = [i for n, i in enumerate(idx) if n == 0 or int(rows[i]['Start_Timestamp']) - int(rows[idx[n - 1]]['Start_Timestamp']) > 200e6]
a = starts[-1]
seg = rows[a:]
t0 = int(seg[0]['Start_Timestamp'])
t1 = max(int(r['End_Timestamp']) for r in seg)
ev = sorted([(int(r['Start_Timestamp']), 1) for r in seg] + [(int(r['End_Timestamp']), -1) for r in seg])
busy = 0
depth = 0
last = t0
hist = {}
for t, d in ev:
    if depth > 0:
        busy += t - last
    hist[depth] = hist.get(depth, 0) + (t - last)
    last = t
    depth += d
k3 = [r for r in seg if 'k_jacobi_solve' in r['Kernel_Name']]
k3_time = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in k3)
cu_avail = sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * min(256, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) for r in k3)
print('last fit: %d kernels over %.1f ms; at least one kernel running %.1f ms (%.0f %%)' % (len(seg), (t1 - t0) / 1e6, busy / 1e6, 100. * busy / (t1 - t0)))
print('kernels in flight -> ms: %s' % {k: round(v / 1e6, 1) for k, v in sorted(hist.items())})
print('K3: %d launches, %.1f ms summed, %.1f CU-s occupied (min(systems,256) CUs x duration; launches of concurrent pipelines share the CUs)'
      % (len(k3), k3_time / 1e6, cu_avail / 1e9))
sizes = {}
for r in k3:
    b = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
    key = '<=32' if b <= 32 else '<=128' if b <= 128 else '<=256' if b <= 256 else '<=1024' if b <= 1024 else '>1024'
    q = sizes.setdefault(key, [0, 0.])
    q[0] += 1
    q[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
print('K3 launches by size (systems): %s' % {k: (v[0], round(v[1], 1)) for k, v in sizes.items()})
