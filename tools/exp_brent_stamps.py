"""Where k_brent_warm spends its cycles (stamped diagnostic build tools/microbench/libvinterp_stamps.so: vi_brent.hip compiled
with -DVI_STAMPS, the rest as shipped): cycle sums of thread 0 over all workgroups, per part, for a batch of T records.
    VINTERP_LIB=tools/microbench/libvinterp_stamps.so python tools/exp_brent_stamps.py [T]"""
import ctypes as C
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('VINTERP_LIB', os.path.join(ROOT, 'tools', 'microbench', 'libvinterp_stamps.so'))
os.environ.setdefault('VINTERP_PIPELINES', '1')
from volumetricinterp_amd import synth, _lib                      # noqa: E402
from volumetricinterp_amd.fitengine import FitEngine              # noqa: E402
from volumetricinterp_amd.models.sphharmlag import Model          # noqa: E402

CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = Model(io.StringIO(CFG))
ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
P, N = lat.size, m.nbasis
d = [ctx.to_device(a) for a in (lat, lon, alt)]
At = m.basis_device(d[0], d[1], d[2], P, transposed=True)
A = At.download().T
R = m.eval_reg_matricies['curvature']()
value, error = synth.synth_records(A, T, seed0=1000)
eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
eng.upload_records(error**-2., value)
eng.fit_resident([P] * T)
out = (C.c_double * 16)()
_lib.lib.vi_debug_brent_stamps.argtypes = [C.POINTER(C.c_double), C.c_int]
_lib.lib.vi_debug_brent_stamps(out, 1)
hist = (C.c_double * 128)()
_lib.lib.vi_debug_brent_hist.argtypes = [C.POINTER(C.c_double), C.c_int]
_lib.lib.vi_debug_brent_hist(hist, 1)
eng.stats = dict(solves=0, launches=0)
t0 = time.perf_counter()
eng.fit_resident([P] * T)
ctx.sync()
t1 = time.perf_counter()
_lib.lib.vi_debug_brent_stamps(out, 0)
v = np.array(list(out))
names = ['queue + first Brent step', 'form X (2 passes over D1, D2)', 'Jacobi solve', 'C = V c', 'chi^2 (A C - b over P points)',
         'Brent step (lane 0)', 'rebase: eigenvector strips', 'rebase: 5 products + yt']
print('T=%d: fit %.1f ms, %d warm solves, %d re-basings' % (T, (t1 - t0) * 1e3, eng.stats.get('warm_solves', 0),
                                                          eng.stats.get('rebased', 0)))
print('   sweeps per solve %.2f, rounds per solve %.1f' % (v[8] / v[9], v[10] / v[9]))
print('   workgroups busy %.1f %% of the launch (sum of their lifetimes / (256 x %.1f ms, first start to last end))' % (100. * v[11] / (256 * (out[13] - out[12])), (out[13] - out[12]) / 1e5))
v = v[:8]
for n_, c_ in zip(names, v):
    print('   %-36s %14.0f cycles  %5.1f %%' % (n_, c_, 100. * c_ / v.sum()))
_lib.lib.vi_debug_brent_hist(hist, 0)
hh = np.array(list(hist)).reshape(64, 2)
print('   sweeps by iteration (iteration: solves, mean sweeps):')
print('   ' + '  '.join('%d: %d, %.1f' % (i, hh[i, 1], hh[i, 0] / max(1., hh[i, 1])) for i in range(64) if hh[i, 1] > 0))
