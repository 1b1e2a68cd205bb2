"""Digest of a rocprofv3 --kernel-trace CSV of bench.py: the launches of the LAST timed step in order (long ones listed,
the rest summed by kernel), so that the latency chain of a single-record fit can be read off.
Usage: python tools/trace_step.py kernel_trace.csv > profiles/rN_bench_kernel_trace.txt"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def nm(r):
    m = re.search(r'(k_\w+(<[^>]*>)?|Cijk_\w{0,34}|__amd\w+)', r['Kernel_Name'])
    return m.group(1) if m else r['Kernel_Name'][:40]


idx = [i for i, r in enumerate(rows) if 'k_scale_rows' in r['Kernel_Name']]       # first kernel of a fit (normal equations)
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
prev, tot, agg = t0, 0, {}
print('one step of bench.py (between two normal-equation passes): launches of 100 us or more, in order')
print('%10s %9s %10s  %s' % ('start us', 'gap us', 'dur us', 'kernel (grid x)'))
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = nm(r)
    if e - s >= 100e3:
        print('%10.1f %9.1f %10.1f  %s (%s)' % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n, r['Grid_Size_X']))
    q = agg.setdefault(n, [0, 0., 0.])
    q[0] += 1
    q[1] += (e - s) / 1e3
    q[2] += max(0, s - prev) / 1e3
    prev = e
    tot += e - s
print('\nkernel time %.1f us of a span of %.1f us (the rest: host between dependent launches)' % (tot / 1e3, (prev - t0) / 1e3))
print('%-44s %5s %12s %14s' % ('kernel', 'calls', 'total us', 'idle before us'))
for n, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-44s %5d %12.1f %14.1f' % (n, c, d, g))
