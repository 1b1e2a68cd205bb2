"""Diagnostic (STAMPS=1 build): rotations per sweep of the solves of a single-record fit (workgroup 0 of each launch)."""
import ctypes as C, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from volumetricinterp_amd import _lib, synth
from volumetricinterp_amd import fitengine
from volumetricinterp_amd.models.sphharmlag import Model
from tools.perf_fit import CFG

real = _lib.lib
def wrap(name):
    f = getattr(real, name)
    def g(*a):
        rc = f(*a)
        ctx.sync()
        out = (C.c_int * 64)()
        real.vi_debug_jacobi_rot(out)
        cnt = []
        for v in out:
            if v < 0: break
            cnt.append(v)
        print('%-26s B=%-4s rotations per sweep (workgroup 0): %s' % (name, a[1], cnt), flush=True)
        return rc
    g.argtypes = f.argtypes
    return g

m = Model(io.StringIO(CFG % (4, 6))); ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
P, N = lat.size, m.nbasis
d = [ctx.to_device(a) for a in (lat, lon, alt)]
At = m.basis_device(d[0], d[1], d[2], P, transposed=True)
A = At.download().T
R = m.eval_reg_matricies['curvature']()
value, error = synth.synth_records(A, 1, seed0=1000)
eng = fitengine.FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
eng.upload_records(error**-2., value)
class L(object):
    def __getattr__(self, n):
        if n in ('vi_solve_trunc_f64', 'vi_warm_prepare_f64', 'vi_warm_solve_f64', 'vi_warm_chi2_one_f64'):
            return wrap(n)
        return getattr(real, n)
fitengine._lib.lib = L()
try:
    eng.fit_resident([P], calccov=True)
finally:
    fitengine._lib.lib = real
