import io, os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from volumetricinterp_amd import synth, _lib
from volumetricinterp_amd.fitengine import FitEngine
from volumetricinterp_amd.models.sphharmlag import Model
CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
T=1000
m = Model(io.StringIO(CFG)); ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
P, N = lat.size, m.nbasis
d = [ctx.to_device(a) for a in (lat, lon, alt)]
At = m.basis_device(d[0], d[1], d[2], P, transposed=True)
A = At.download().T
R = m.eval_reg_matricies['curvature']()
value, error = synth.synth_records(A, T, seed0=1000)
eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
eng.upload_records(error**-2., value)
eng.fit_resident([P]*T)
eng.stats = dict(solves=0, launches=0)
t0=time.perf_counter(); res=eng.fit_resident([P]*T); ctx.sync(); t1=time.perf_counter()
print((t1-t0)*1e3, 'ms'); print(eng.stats)
inf = res['search']['curvature']
its=[i.get('iterations',0) for i in inf['info'] if i]
print('info keys', inf['info'][0].keys() if inf['info'][0] else None)
a=np.array(its); print('brent iterations: mean %.1f max %d; >20: %d; >40: %d'%(a.mean(), a.max(), (a>20).sum(), (a>40).sum()))
lo=[i.get('bracket',(0,0))[0] for i in inf['info'] if i and 'bracket' in i]
print('bracket lower ends: ', np.unique(np.array(lo), return_counts=True))
I = inf['info']
oc = inf['outcomes']
print('guard: jump %d, polished_cold %d, redone_cold %d, inconsistent %d of %d roots' % (
    sum(1 for i in I if i and i.get('jump')), len(inf.get('polished_cold', [])), len(inf.get('redone_cold', [])),
    sum(1 for i, o in zip(I, oc) if o == 'root' and not i.get('consistent')), oc.count('root')))
print('polish iterations:', [i.get('polish_iterations') for i in I if i and i.get('polished_cold')])
os.environ['VINTERP_STAGE_TIMES'] = '1'
eng.stats = dict(solves=0, launches=0)
res = eng.fit_resident([P]*T); ctx.sync()
print({k: round(v, 1) for k, v in eng.stats.items() if k.startswith('ms_')})
