#!/usr/bin/env python3
"""Wall time of the user-facing Estimate.__call__ (host arrays in, host array out) at 128^3."""
import os, sys, time, datetime as dt
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_golden
from volumetricinterp_amd import synth
from volumetricinterp_amd.estimate import Estimate
f = load_golden('fit_default')
es = Estimate.from_arrays(f['Coeffs'], f['Covariance'], f['utime'], f['hull_vert'], str(f['cfg']))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = synth.query_grid(n)
t = dt.datetime(1970, 1, 1) + dt.timedelta(seconds=float(np.mean(f['utime'][0])))
for hull in (False, True):
    es(t, *g, check_hull=hull)
    t0 = time.perf_counter()
    for _ in range(5):
        out = es(t, *g, check_hull=hull)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print('Estimate.__call__ n=%d check_hull=%s: %.2f ms wall -> %.3e points/s (host in/out, %d MB moved)'
          % (n, hull, ms, g[0].size / ms * 1e3, g[0].size * 32 // 2**20))
