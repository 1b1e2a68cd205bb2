"""Host-side profile (cProfile) of single-record fits: 20 fits of the bench record."""
import cProfile, io, os, pstats, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth
from volumetricinterp_amd.fitengine import FitEngine
from volumetricinterp_amd.models.sphharmlag import Model
CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
m = Model(io.StringIO(CFG)); ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
P, N = lat.size, m.nbasis
d = [ctx.to_device(a) for a in (lat, lon, alt)]
At = m.basis_device(d[0], d[1], d[2], P, transposed=True)
A = At.download().T
R = m.eval_reg_matricies['curvature']()
value, error = synth.synth_records(A, 1, seed0=1000)
eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
W = error**-2.
for _ in range(3):
    eng.upload_records(W, value); eng.fit_resident([P], calccov=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    eng.upload_records(W, value)
    eng.fit_resident([P], calccov=True)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue())
