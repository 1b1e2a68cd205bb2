"""Round by round through the fit of one pipeline's share of workload c3: T records (default 2500) in ONE pipeline with
VINTERP_TRACE=1 (one line per search round: systems by kind, wall time) and VINTERP_STAGE_TIMES=1, plus the solve-timing
counters (K3 launches, systems, rounds).  python tools/trace_batch.py [T]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['VINTERP_PIPELINES'] = '1'
from volumetricinterp_amd import synth                                   # noqa: E402
from test_gpu_configs import _engine, CFG144                              # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
m, ctx, eng, A, _ = _engine(CFG144, synth.GEOM_C2)
P = A.shape[0]
value, error = synth.synth_records(A, T, seed0=1000)
eng.upload_records(error**-2., value)
eng.fit_resident([P] * T)                                                   # warm-up
os.environ['VINTERP_STAGE_TIMES'] = '1'
os.environ['VINTERP_TRACE'] = '1'
for k in list(eng.stats):
    if k.startswith('ms_'):
        del eng.stats[k]
eng.solve_timing(1)
ctx.sync()
t0 = time.perf_counter()
res = eng.fit_resident([P] * T)
ctx.sync()
wall = time.perf_counter() - t0
st = eng.solve_timing(0)
print('T = %d in one pipeline: %.1f ms = %.0f records/s;' % (T, wall * 1e3, T / wall),
      {k: round(v, 1) for k, v in eng.stats.items() if k.startswith('ms_')})
print('K3 launches %d, systems %d, rounds %d (%.1f sweeps per record), launch ms summed %.1f'
      % (st['launches'], st['systems'], st['rounds'], st['rounds'] / 72. / T, st['total_ms'] * st['launches'] / max(1, st['timed'])))
