"""Diagnostic: alpha / chi^2 of records fitted alone, in a batch of 8 and in a batch of 40 (the comparison of
tests/test_gpu_configs.py::test_c1_fit_is_independent_of_the_batch_and_consistent), printed pair by pair."""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_configs import _engine, CFG144
from volumetricinterp_amd import synth
m, ctx, eng, A, _ = _engine(CFG144, synth.GEOM_C2)
P = A.shape[0]
value, error = synth.synth_records(A, 40, seed0=1000)
W = error**-2.
full = eng.fit(W, value, [P] * 40)
eight = eng.fit(W[:8], value[:8], [P] * 8)
for t in range(8):
    one = eng.fit(W[t:t + 1], value[t:t + 1], [P])
    a1 = one['reg_params'][0]['curvature']
    i1 = one['search']['curvature']['info'][0]
    for nm, other in (('8', eight), ('40', full)):
        a2 = other['reg_params'][t]['curvature']
        io = other['search']['curvature']['info'][t]
        if np.isnan(a1) or np.isnan(a2):
            print(t, nm, a1, a2); continue
        print('rec %d vs batch %2s: dlog10a %.2e  dchi2/chi2 %.2e  it %s/%s finder %s/%s bracket %s/%s' % (
            t, nm, abs(math.log10(a1) - math.log10(a2)), abs(one['chi_sq'][0] - other['chi_sq'][t]) / one['chi_sq'][0],
            i1.get('iterations'), io.get('iterations'), i1.get('finder'), io.get('finder'), i1.get('bracket'), io.get('bracket')))
