"""One rank's share of configs[3] at N = 8 (1250 records) fitted with 1, 2, 3, 4 pipelines, and the stages of one 312-record
pipeline: where the fixed costs of a small batch are.   python tools/exp_small_batch.py [T]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from volumetricinterp_amd import synth                                   # noqa: E402
from test_gpu_configs import _engine, CFG144                              # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
m, ctx, eng, A, _ = _engine(CFG144, synth.GEOM_C2)
P = A.shape[0]
value, error = synth.synth_records(A, T, seed0=1000)
for pipes in ('4', '1', '2', '3', '4', '6'):
    os.environ['VINTERP_PIPELINES'] = pipes
    eng.upload_records(error**-2., value)
    eng.fit_resident([P] * T)
    ctx.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        eng.fit_resident([P] * T)
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    print('T = %d, %s pipelines: %.1f ms = %.0f records/s' % (T, pipes, best * 1e3, T / best))
os.environ['VINTERP_PIPELINES'] = '1'
os.environ['VINTERP_STAGE_TIMES'] = '1'
os.environ['VINTERP_TRACE'] = '1'
n = T // 4
eng.upload_records(error[:n]**-2., value[:n])
eng.fit_resident([P] * n)
for k in list(eng.stats):
    if k.startswith('ms_'):
        del eng.stats[k]
t0 = time.perf_counter()
eng.fit_resident([P] * n)
ctx.sync()
print('T = %d in one pipeline: %.1f ms;' % (n, (time.perf_counter() - t0) * 1e3), {k: round(v, 1) for k, v in eng.stats.items() if k.startswith('ms_')})
