"""Per-stream timeline of the LAST batched fit in a rocprofv3 results database (rocprofv3 --kernel-trace): for every stream
its kernel count, busy time, first start and last end; for one stream (default: the busiest) the launches of --min-us or
more, and gaps of that size, in order.    python tools/trace_streams.py x_results.db [--stream K] [--min-us 500]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.cursor().execute('select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start').fetchall()
arg = lambda k, d: type(d)(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d      # noqa: E731
min_us = arg('--min-us', 500.)


def nm(n):
    m = re.search(r'(k_\w+|Cijk_\w{0,12}|__amd\w+)', n)
    return m.group(1) if m else n[:30]


idx = [i for i, r in enumerate(rows) if 'k_scale_rows' in r[0]]
starts = [i for n, i in enumerate(idx) if n == 0 or rows[i][1] - rows[idx[n - 1]][1] > 100e6]
seg = rows[starts[-1]:]
t0, t1 = seg[0][1], max(r[2] for r in seg)
print('last fit: span %.1f ms, %d kernels' % ((t1 - t0) / 1e6, len(seg)))
streams = sorted(set(r[3] for r in seg))
busy = {}
for s in streams:
    ss = [r for r in seg if r[3] == s]
    busy[s] = sum(r[2] - r[1] for r in ss) / 1e6
    print('stream %d: %4d kernels, busy %.1f ms, first start %.1f ms, last end %.1f ms' % (s, len(ss), busy[s], (ss[0][1] - t0) / 1e6,
                                                                                       (ss[-1][2] - t0) / 1e6))
s = arg('--stream', max(busy, key=busy.get))
ss = [r for r in seg if r[3] == s]
prev = ss[0][1]
print('stream %d: launches / gaps of %.0f us or more' % (s, min_us))
small_n, small_t = 0, 0.
for r in ss:
    d, gap = (r[2] - r[1]) / 1e3, (r[1] - prev) / 1e3
    if d >= min_us or gap >= min_us:
        if small_n:
            print('%10s  ... %d shorter launches, %.2f ms' % ('', small_n, small_t / 1e3))
            small_n, small_t = 0, 0.
        print('%8.1f ms  gap %7.2f ms  %8.2f ms  %s (%d workgroups)' % ((r[1] - t0) / 1e6, gap / 1e3, d / 1e3, nm(r[0]), r[4] // max(1, r[5])))
    else:
        small_n += 1
        small_t += d
    prev = max(prev, r[2])
if small_n:
    print('%10s  ... %d shorter launches, %.2f ms' % ('', small_n, small_t / 1e3))
