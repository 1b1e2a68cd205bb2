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

# the same call with the grid and the output in page-locked memory (coordinates up and densities down overlap fully)
from volumetricinterp_amd import _lib
pg = [_lib.pinned_empty(a.shape) for a in g]
for d, a in zip(pg, g):
    d[...] = a
po = _lib.pinned_empty((1, g[0].size))
C, _ = es.get_C(t)
for hull in (False, True):
    es.evaluate_coeffs(C[None, :], *pg, check_hull=hull, out=po)
    t0 = time.perf_counter()
    for _ in range(5):
        es.evaluate_coeffs(C[None, :], *pg, check_hull=hull, out=po)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print('evaluate_coeffs, page-locked arrays n=%d check_hull=%s: %.2f ms wall -> %.3e points/s'
          % (n, hull, ms, g[0].size / ms * 1e3))
ref = es(t, *g, check_hull=True)
assert np.array_equal(np.isnan(ref.ravel()), np.isnan(po[0])) and np.allclose(np.nan_to_num(ref.ravel()), np.nan_to_num(po[0]), rtol=0, atol=0)
print('pinned and pageable paths agree bit for bit')
