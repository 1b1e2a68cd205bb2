"""Statistics of the consistency guard on a big batch: how the records of a T-record fit ended (consistent / within the
band / jump / polished / redone), and the largest |chi^2 - nu| / nu among the records that are not flagged as jumps.
Usage (GPU): python tools/stats_guard.py [T]"""
import io, os, sys, time, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth
from volumetricinterp_amd.fitengine import FitEngine
from volumetricinterp_amd.models.sphharmlag import Model
CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
m = Model(io.StringIO(CFG)); ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
P, N = lat.size, m.nbasis
d = [ctx.to_device(a) for a in (lat, lon, alt)]
At = m.basis_device(d[0], d[1], d[2], P, transposed=True)
A = At.download().T
R = m.eval_reg_matricies['curvature']()
value, error = synth.synth_records(A, T, seed0=20000)
eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
t0 = time.perf_counter(); r = eng.fit(error**-2., value, [P] * T); t1 = time.perf_counter()
inf = r['search']['curvature']
cnt = collections.Counter(); worst = 0.; its = []
for t in range(T):
    o = inf['outcomes'][t]; i = inf['info'][t]
    if o != 'root':
        cnt[o] += 1; continue
    nu = i['sf'] * P
    miss = abs(r['chi_sq'][t] - nu) / nu
    its.append(i.get('iterations', 0))
    if i.get('redone_cold'): cnt['redone_cold'] += 1
    elif i.get('polished_cold'): cnt['polished (%s)' % i.get('polish_end')] += 1
    elif i.get('jump'): cnt['jump'] += 1
    elif i.get('consistent'): cnt['consistent (<= 1e-6 nu)'] += 1
    else: cnt['within 1e-4 nu'] += 1
    if not i.get('jump'):
        worst = max(worst, miss)
    if i.get('walk_redone_exact'): cnt['walk redone exact'] += 1
print('T=%d (first call, %.2f s): %s' % (T, t1 - t0, dict(cnt)))
mv = np.array([abs(i['log10_alpha'] - i['warm_log10_alpha']) for i in inf['info'] if i.get('polished_cold')])
if len(mv):
    print('polished records: |log10 alpha - warm root| quantiles 10/50/90/100 %%: %s; polish rounds: median %d max %d' % (
        np.quantile(mv, [0.1, 0.5, 0.9, 1.0]).tolist(), np.median([i['polish_iterations'] for i in inf['info'] if i.get('polished_cold')]),
        max(i['polish_iterations'] for i in inf['info'] if i.get('polished_cold'))))
print('largest |chi2 - nu| / nu among records not flagged as jumps: %.2e; Brent iterations: median %d, 90%% %d, max %d; stats %s'
      % (worst, np.median(its), np.quantile(its, 0.9), max(its), eng.stats))
