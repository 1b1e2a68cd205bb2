import io, os, sys, time, collections
import numpy as np
sys.path.insert(0, '/root/repo')
os.environ['VINTERP_PIPELINES']='1'
from volumetricinterp_amd import synth
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
r=eng.fit_resident([P]*T)
inf=r['search']['curvature']
c=collections.Counter()
dev=[]
for t in range(T):
    if inf['outcomes'][t]!='root': c[inf['outcomes'][t]]+=1; continue
    i=inf['info'][t]
    k='consistent' if i.get('consistent') else ('jump' if i.get('jump') else ('redone' if i.get('redone_cold') else ('polished' if t in inf.get('polished_cold',[]) else 'band')))
    c[k]+=1
    dev.append(abs(i['chi2_minus_nu'])/ (i['sf']*P))
print(c, 'polished', len(inf.get('polished_cold',[])), 'redone', len(inf.get('redone_cold',[])))
dev=np.array(dev); print('chi2 miss / nu quantiles', np.quantile(dev,[.5,.9,.99,1.0]))
its=[inf['info'][t].get('iterations') for t in range(T) if inf['outcomes'][t]=='root']
print('iterations quantiles', np.quantile(its,[.5,.85,.9,.99,1.0]), 'mean', np.mean(its))
