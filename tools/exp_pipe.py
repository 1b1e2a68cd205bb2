"""Timing of FitEngine.fit_resident with 1 .. 4 pipelines (VINTERP_PIPELINES) and equality of the answers."""
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
value, error = synth.synth_records(A, T, seed0=1000)
ref = None
for K in (sys.argv[2:] or ['1', '2', '3', '4', 'auto']):
    if K == 'auto':
        os.environ.pop('VINTERP_PIPELINES', None)
    else:
        os.environ['VINTERP_PIPELINES'] = K
    eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
    eng.upload_records(error**-2., value)
    eng.fit_resident([P] * T, calccov=True)
    eng.stats = dict(solves=0, launches=0)
    t0 = time.perf_counter(); r = eng.fit_resident([P] * T, calccov=True); ctx.sync(); t1 = time.perf_counter()
    if ref is None:
        ref = r
    same = all(np.array_equal(r[k], ref[k], equal_nan=True) for k in ('Coeffs', 'chi_sq', 'Covariance'))
    same_a = all((a['curvature'] == b['curvature']) or (np.isnan(a['curvature']) and np.isnan(b['curvature']))
                 for a, b in zip(r['reg_params'], ref['reg_params']))
    print('pipelines=%s (%s): %.1f ms -> %.1f records/s; alpha identical: %s, Coeffs/chi2/Cov identical: %s; stats %s'
          % (K, eng.stats.get('pipelines', 1), (t1 - t0) * 1e3, T / (t1 - t0), same_a, same, eng.stats), flush=True)
    eng.close()
