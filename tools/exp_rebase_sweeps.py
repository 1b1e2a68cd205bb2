"""Diagnostic: Jacobi sweeps of a warm solve as a function of the distance (in decades) between alpha and the alpha the
rotated system was set up at - what would re-basing the rotated system near the root buy Brent's late iterates?"""
import io, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from volumetricinterp_amd import _lib, fitengine, synth
from volumetricinterp_amd.models.sphharmlag import Model
CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
m = Model(io.StringIO(CFG)); ctx = m.ctx; h = ctx.handle
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
P, N = lat.size, m.nbasis
d = [ctx.to_device(a) for a in (lat, lon, alt)]
A = m.basis_device(d[0], d[1], d[2], P, transposed=True).download().T
R = m.eval_reg_matricies['curvature']()
value, error = synth.synth_records(A, 4, seed0=1000)
W = error**-2.
EPS = np.finfo(float).eps
ctx.solve_timing(1)
for j, x0 in ((0, -26.37), (1, -25.4), (3, -26.6)):
    AWA = (A.T * W[j]) @ A; y = A.T @ (W[j] * value[j])
    dAWA = ctx.to_device(AWA[None]); dy = ctx.to_device(y[None]); dR = ctx.to_device(R)
    rec = ctx.to_device(np.zeros(1, np.int32))
    dV = ctx.empty((1, N, N)); dD1 = ctx.empty((1, N, N)); dD2 = ctx.empty((1, N, N)); dyt = ctx.empty((1, N)); dC0 = ctx.empty((1, N))
    drk = ctx.empty((1,), np.int32); dC = ctx.empty((1, N))
    da0 = ctx.to_device(np.array([10.0**x0]))
    _lib.check(_lib.lib.vi_warm_prepare_f64(h, 1, N, dAWA.ptr, rec.ptr, da0.ptr, dR.ptr, dy.ptr, EPS,
                                            dC0.ptr, drk.ptr, dV.ptr, dD1.ptr, dD2.ptr, dyt.ptr), 'prep')
    row = []
    for dx in (0.5, 0.2, 0.1, 3e-2, 1e-2, 1e-3, 1e-4, 1e-6, 1e-9, 0.):
        da = ctx.to_device(np.array([10.0**(x0 + dx)]))
        ctx.sync(); ctx.solve_timing(1)
        ctx.timer_start()
        _lib.check(_lib.lib.vi_warm_solve_f64(h, 1, N, dD1.ptr, dD2.ptr, dyt.ptr, dV.ptr, rec.ptr, da.ptr, EPS, dC.ptr, drk.ptr, None), 'warm')
        ms = ctx.timer_stop_ms()
        row.append('%g: %.1f sw %.2f ms' % (dx, ctx.solve_timing(1)['rounds'] / 72., ms))
    print('rec %d basis at %.2f | ' % (j, x0) + ' | '.join(row))
