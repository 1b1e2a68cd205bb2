"""A/B of the bracket walk in shared bases (VINTERP_SHAREDWALK): time and answers of a batched fit, both ways
(MODES=c1: "c" or "0" = every walk system solved cold, "1" = shared bases).
Usage (GPU): python tools/exp_sharedwalk_ab.py [T]"""
import io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth
from volumetricinterp_amd.fitengine import FitEngine
from volumetricinterp_amd.models.sphharmlag import Model
CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = Model(io.StringIO(CFG)); ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
P, N = lat.size, m.nbasis
d = [ctx.to_device(a) for a in (lat, lon, alt)]
At = m.basis_device(d[0], d[1], d[2], P, transposed=True)
A = At.download().T
R = m.eval_reg_matricies['curvature']()
value, error = synth.synth_records(A, T, seed0=1000)
res = {}
for mode in tuple(os.environ.get('MODES', '01')):
    os.environ['VINTERP_SHAREDWALK'] = '1' if mode == '1' else '0'
    eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
    eng.upload_records(error**-2., value)
    eng.fit_resident([P] * T, calccov=True)
    eng.stats = dict(solves=0, launches=0)
    ctx.solve_timing(1)
    acct = {}
    if os.environ.get('ACCT') == '1':
        orig = eng.chi2_batch_search

        def wrapped(rec, log10a, name, exact=None, _o=orig):
            ctx.sync(); r0 = ctx.solve_timing(1)['rounds']; ta = time.perf_counter()
            out = _o(rec, log10a, name, exact)
            ctx.sync(); tb = time.perf_counter(); r1 = ctx.solve_timing(1)['rounds']
            li = np.asarray(log10a)
            kind = 'walk' if np.all(li == np.floor(li)) else ('brent' if not np.any(li == np.floor(li)) else 'mixed')
            a = acct.setdefault(kind, [0, 0, 0., 0])
            a[0] += len(rec); a[1] += r1; a[2] += tb - ta; a[3] += 1
            return out
        eng.chi2_batch_search = wrapped
        origf = eng.finalize

        def wrappedf(*a, **k):
            ctx.sync(); ctx.solve_timing(1); ta = time.perf_counter()
            out = origf(*a, **k)
            ctx.sync(); tb = time.perf_counter(); r1 = ctx.solve_timing(1)['rounds']
            q = acct.setdefault('final', [0, 0, 0., 0]); q[0] += T; q[1] += r1; q[2] += tb - ta; q[3] += 1
            return out
        eng.finalize = wrappedf
        for nm in ('search', 'chi2_batch', 'form_normal_equations'):
            def mk(nm, f):
                def w(*a, **k):
                    ctx.sync(); ta = time.perf_counter()
                    out = f(*a, **k)
                    ctx.sync(); q = acct.setdefault('fn:' + nm, [0, 0, 0., 0]); q[2] += time.perf_counter() - ta; q[3] += 1
                    return out
                return w
            setattr(eng, nm, mk(nm, getattr(eng, nm)))
    t0 = time.perf_counter(); r = eng.fit_resident([P] * T, calccov=True); ctx.sync(); t1 = time.perf_counter()
    for kind, (n, rd, tt, calls) in acct.items():
        print('   %-6s %7d systems in %3d calls: %6.1f sweeps/record, %7.1f ms' % (kind, n, calls, rd / 72. / T, tt * 1e3))
    st = ctx.solve_timing(0)
    inf = r['search']['curvature']
    oc = inf['outcomes']
    print('shared=%s T=%d: %.1f ms -> %.1f records/s; stats %s; K3 %.1f ms in %d launches, %d rounds; outcomes %s; polished %d redone %d'
          % (mode, T, (t1 - t0) * 1e3, T / (t1 - t0), eng.stats, st['total_ms'], st['launches'], st['rounds'],
             {o: oc.count(o) for o in set(oc)}, len(inf.get('polished_cold', [])), len(inf.get('redone_cold', []))))
    res[mode] = r
    eng.close()
if len(res) < 2:
    sys.exit(0)
k0, k1 = list(res)[:2]
a0 = np.array([p['curvature'] for p in res[k0]['reg_params']]); a1 = np.array([p['curvature'] for p in res[k1]['reg_params']])
dl = np.abs(np.log10(a1) - np.log10(a0))
print('log10 alpha: max |diff| %.2e, records over 1e-6: %d, over 1e-3: %d' % (np.nanmax(dl), np.sum(dl > 1e-6), np.sum(dl > 1e-3)))
c0, c1 = res[k0]['chi_sq'], res[k1]['chi_sq']
print('chi^2: max rel diff %.2e' % np.nanmax(np.abs(c1 - c0) / c0))
o0, o1 = res[k0]['search']['curvature']['outcomes'], res[k1]['search']['curvature']['outcomes']
print('outcome flips: %d' % sum(x != y for x, y in zip(o0, o1)))
sf0 = [i.get('sf') for i in res[k0]['search']['curvature']['info']]; sf1 = [i.get('sf') for i in res[k1]['search']['curvature']['info']]
print('records whose coefficients differ at all: %d' % int(np.sum([not np.array_equal(x, y, equal_nan=True) for x, y in zip(res[k0]['Coeffs'], res[k1]['Coeffs'])])))
print('walks redone exact (shared): %d' % sum(1 for i in res[k1]['search']['curvature']['info'] if i.get('walk_redone_exact')))
print('scale factor flips: %d' % sum(x != y for x, y in zip(sf0, sf1)))
