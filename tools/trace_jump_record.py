"""Where the 42 ms of the bench's jump record (record 14 of bench.py's sixteen, seed 1014) go: the record fitted alone with the
Python-driven loop and VINTERP_TRACE=1 (one line per search round with its wall time), then stage times of the default path.
python tools/trace_jump_record.py [record]"""
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from volumetricinterp_amd import synth                                   # noqa: E402
from test_gpu_configs import _engine, CFG144                              # noqa: E402

t = int(sys.argv[1]) if len(sys.argv) > 1 else 14
m, ctx, eng, A, _ = _engine(CFG144, synth.GEOM_C2)
P = A.shape[0]
value, error = synth.synth_records(A, 16, seed0=1000)
W = error**-2.
eng.fit(W[t:t + 1], value[t:t + 1], [P])                                   # warm-up
os.environ['VINTERP_STAGE_TIMES'] = '1'
for k in list(eng.stats):
    if k.startswith('ms_'):
        del eng.stats[k]
t0 = time.perf_counter()
res = eng.fit(W[t:t + 1], value[t:t + 1], [P])
print('default path: %.2f ms;' % ((time.perf_counter() - t0) * 1e3), {k: round(v, 2) for k, v in eng.stats.items() if k.startswith('ms_')})
print(res['search']['curvature']['info'][0])
os.environ['VINTERP_HOST_LOOP_BRENT'] = '0'
os.environ['VINTERP_TRACE'] = '1'
t0 = time.perf_counter()
res = eng.fit(W[t:t + 1], value[t:t + 1], [P])
print('python-driven loop: %.2f ms' % ((time.perf_counter() - t0) * 1e3))
