"""Per-kernel summary of a rocprofv3 results database (rocprofv3 --kernel-trace -d DIR -o NAME writes DIR/NAME_results.db
on this image): calls, total / mean / min / max duration.  With --seq N also the last N dispatches in order.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--seq 40] [--csv out.csv]"""
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
