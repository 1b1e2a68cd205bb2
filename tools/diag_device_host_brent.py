"""Brent's iteration on the device against the host-driven one on a big batch: T fresh default-order records fitted twice,
alpha, chi^2, coefficients, covariances and iteration counts compared bit for bit.   python tools/diag_device_host_brent.py [T]"""
import io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth
from volumetricinterp_amd.fitengine import FitEngine
from volumetricinterp_amd.models.sphharmlag import Model
CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = Model(io.StringIO(CFG)); ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
P, N = lat.size, m.nbasis
d = [ctx.to_device(a) for a in (lat, lon, alt)]
At = m.basis_device(d[0], d[1], d[2], P, transposed=True)
A = At.download().T
R = m.eval_reg_matricies['curvature']()
value, error = synth.synth_records(A, T, seed0=31000)
eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
res = {}
for mode in ('1', '0'):
    os.environ['VINTERP_DEVICE_BRENT'] = mode
    t0 = time.perf_counter(); res[mode] = eng.fit(error**-2., value, [P] * T); t1 = time.perf_counter()
    print('VINTERP_DEVICE_BRENT=%s: %.1f ms' % (mode, (t1 - t0) * 1e3))
a, b = res['1'], res['0']
same = {k: bool(np.array_equal(a[k], b[k], equal_nan=True)) for k in ('Coeffs', 'Covariance', 'chi_sq', 'ranks')}
al = [(x['curvature'], y['curvature']) for x, y in zip(a['reg_params'], b['reg_params'])]
same['alpha'] = all(x == y or (np.isnan(x) and np.isnan(y)) for x, y in al)
ia, ib = a['search']['curvature']['info'], b['search']['curvature']['info']
same['iterations'] = [i.get('iterations') for i in ia] == [i.get('iterations') for i in ib]
same['outcomes'] = a['search']['curvature']['outcomes'] == b['search']['curvature']['outcomes']
print('identical:', same)
sys.exit(0 if all(same.values()) else 1)
