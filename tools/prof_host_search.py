"""cProfile of one 1000-record fit in one pipeline (host side of the batched search).  python tools/prof_host_search.py"""
import cProfile
import io
import os
import pstats
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['VINTERP_PIPELINES'] = '1'
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
eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
eng.upload_records(error**-2., value)
eng.fit_resident([P] * T, calccov=True)
pr = cProfile.Profile()
pr.enable()
eng.fit_resident([P] * T, calccov=True)
ctx.sync()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue())
