"""A record fitted alone (host-driven path, Brent's loop in C) against the same record inside a batch (shared-basis walk, Brent's
iteration on the device): T fresh default-order records, alpha / chi^2 / coefficients / covariance compared bit for bit.
python tools/diag_alone_vs_batch.py [T]"""
import io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth
from volumetricinterp_amd.fitengine import FitEngine
from volumetricinterp_amd.models.sphharmlag import Model
CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = Model(io.StringIO(CFG)); ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
P, N = lat.size, m.nbasis
d = [ctx.to_device(a) for a in (lat, lon, alt)]
At = m.basis_device(d[0], d[1], d[2], P, transposed=True)
A = At.download().T
R = m.eval_reg_matricies['curvature']()
value, error = synth.synth_records(A, T, seed0=47000)
W = error**-2.
eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
batch = eng.fit(W, value, [P] * T)
bad = []
for t in range(T):
    one = eng.fit(W[t:t + 1], value[t:t + 1], [P])
    ok = all(np.array_equal(one[k][0], batch[k][t], equal_nan=True) for k in ('Coeffs', 'Covariance', 'chi_sq'))
    x, y = one['reg_params'][0]['curvature'], batch['reg_params'][t]['curvature']
    ok = ok and (x == y or (np.isnan(x) and np.isnan(y)))
    if not ok:
        bad.append(t)
its = [i.get('iterations', 0) for i in batch['search']['curvature']['info'] if i]
print('%d records: alone == in batch bit for bit on %d; differing: %s; Brent iterations median %d max %d' %
      (T, T - len(bad), bad, int(np.median(its)), max(its)))
sys.exit(1 if bad else 0)
