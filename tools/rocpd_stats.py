"""Per-kernel summary of a rocprofv3 results database (rocprofv3 --kernel-trace -d DIR -o NAME writes DIR/NAME_results.db
on this image): calls, total / mean / min / max duration.  With --seq N also the last N dispatches in order.
With --step K the dispatches of the K-th last single-record fit step (delimited by k_scale_rows, the first kernel of a
fit's normal equations) in order: launches of 100 us or more listed, the rest summed by kernel - the latency chain.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--seq 40] [--csv out.csv] [--step 1]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute('select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 '
                   'from kernels group by name order by 3 desc').fetchall()
tot = sum(r[2] for r in rows)
lines = ['kernel,calls,total_us,mean_us,min_us,max_us,percent']
for r in rows:
    lines.append('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % (r[0], r[1], r[2], r[3], r[4], r[5], 100. * r[2] / tot))
    print('%-70s n=%5d tot %11.1f us  mean %9.1f  min %9.1f  max %9.1f  %5.1f %%' % (r[0][:70], r[1], r[2], r[3], r[4], r[5],
                                                                                 100. * r[2] / tot))
if '--csv' in sys.argv:
    open(sys.argv[sys.argv.index('--csv') + 1], 'w').write('\n'.join(lines) + '\n')
if '--seq' in sys.argv:
    n = int(sys.argv[sys.argv.index('--seq') + 1])
    seq = cur.execute('select name, start, end, grid_x, workgroup_x from kernels order by start').fetchall()
    t0 = seq[-n][1] if len(seq) >= n else seq[0][1]
    for r in seq[-n:]:
        print('%10.1f  %-60s grid %7d wg %4d  %9.1f us' % ((r[1] - t0) / 1e3, r[0][:60], r[3], r[4], (r[2] - r[1]) / 1e3))

if '--step' in sys.argv:
    kth = int(sys.argv[sys.argv.index('--step') + 1])
    seq = cur.execute('select name, start, end, grid_x from kernels order by start').fetchall()

    def nm(n):
        m = re.search(r'(k_\w+(<[^>]*>)?|Cijk_\w{0,34}|__amd\w+)', n)
        return m.group(1) if m else n[:40]
    idx = [i for i, r in enumerate(seq) if 'k_scale_rows' in r[0]]
    a, b = idx[-kth - 1], idx[-kth]
    t0 = seq[a][1]
    prev, tot, agg = t0, 0., {}
    print('one step (between two normal-equation passes): launches of 100 us or more, in order')
    print('%10s %9s %10s  %s' % ('start us', 'gap us', 'dur us', 'kernel (grid x)'))
    for n, s_, e_, gx in seq[a:b]:
        k = nm(n)
        if e_ - s_ >= 100e3:
            print('%10.1f %9.1f %10.1f  %s (%d)' % ((s_ - t0) / 1e3, (s_ - prev) / 1e3, (e_ - s_) / 1e3, k, gx))
        q = agg.setdefault(k, [0, 0., 0.])
        q[0] += 1
        q[1] += (e_ - s_) / 1e3
        q[2] += max(0., (s_ - prev) / 1e3)
        tot += (e_ - s_) / 1e3
        prev = max(prev, e_)
    span = (seq[b][1] - t0) / 1e3
    print('\nkernel time %.1f us of a span of %.1f us (the rest: host between dependent launches)' % (tot, span))
    print('%-44s %6s %12s %14s' % ('kernel', 'calls', 'total us', 'idle before us'))
    for k, q in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-44s %6d %12.1f %14.1f' % (k, q[0], q[1], q[2]))
