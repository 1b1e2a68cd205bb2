"""Diagnostic: Jacobi sweeps of the bracket-walk systems X(10^k) = AWA + 10^k R, k = 0 .. -45, solved cold and solved in the
eigenbasis of X(10^k0) for several k0 (warm) - is there a small set of bases from which every walk system converges fast?
Usage (GPU): python tools/exp_basis_sweeps.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden, rel
from volumetricinterp_amd import _lib, fitengine
ctx = _lib.get_context(); h = ctx.handle
f = load_golden('fit_default'); R = f['R']; AWA = f['rec0_AWA']; y = f['rec0_y']; N = 144
EPS = np.finfo(float).eps
M = (N + 3) // 4
ROUNDS = 2 * M                                     # rounds per sweep (m units = 2M: 1 intra + m - 1 cross)
dAWA = ctx.to_device(AWA[None]); dy = ctx.to_device(y[None]); dR = ctx.to_device(R)
rec = ctx.to_device(np.zeros(1, np.int32))
ks = np.arange(0, -46, -1.0)
dX = ctx.empty((1, N, N)); dC = ctx.empty((1, N)); drk = ctx.empty((1,), np.int32)


def rounds():
    return ctx.solve_timing(1)['rounds']


ctx.solve_timing(1)


cold, Cc, rkc = [], [], []
for k in ks:
    da = ctx.to_device(np.array([10.0**k]))
    _lib.check(_lib.lib.vi_form_system_f64(h, 1, N, dAWA.ptr, rec.ptr, da.ptr, dR.ptr, dX.ptr), 'form')
    rounds()
    _lib.check(_lib.lib.vi_solve_trunc_f64(h, 1, N, dX.ptr, dy.ptr, rec.ptr, EPS, dC.ptr, drk.ptr, N * EPS, None), 'cold')
    ctx.sync()
    cold.append(rounds() / ROUNDS); Cc.append(dC.download()[0].copy()); rkc.append(int(drk.download()[0]))
print('rounds per sweep %d' % ROUNDS)
print('k     :', ' '.join('%3d' % k for k in ks))
print('rank  :', ' '.join('%3d' % r for r in rkc))
print('cold  :', ' '.join('%3.0f' % s for s in cold))
dV = ctx.empty((1, N, N)); dD1 = ctx.empty((1, N, N)); dD2 = ctx.empty((1, N, N)); dyt = ctx.empty((1, N)); dC0 = ctx.empty((1, N))
best = np.full(len(ks), 99.)
for k0 in (0, -8, -14, -18, -22, -25, -27, -29, -31, -34, -40):
    da0 = ctx.to_device(np.array([10.0**k0]))
    _lib.check(_lib.lib.vi_warm_prepare_f64(h, 1, N, dAWA.ptr, rec.ptr, da0.ptr, dR.ptr, dy.ptr, EPS, dC0.ptr, drk.ptr,
                                            dV.ptr, dD1.ptr, dD2.ptr, dyt.ptr), 'prep')
    sw, err = [], []
    for i, k in enumerate(ks):
        da = ctx.to_device(np.array([10.0**k]))
        ctx.sync(); rounds()
        _lib.check(_lib.lib.vi_warm_solve_f64(h, 1, N, dD1.ptr, dD2.ptr, dyt.ptr, dV.ptr, rec.ptr, da.ptr, EPS, dC.ptr, drk.ptr, None), 'warm')
        ctx.sync()
        sw.append(rounds() / ROUNDS); err.append(rel(dC.download()[0], Cc[i]))
    best = np.minimum(best, sw)
    print('k0=%3d:' % k0, ' '.join('%3.0f' % s for s in sw))
    print('  relC:', ' '.join('%3.0f' % (-np.log10(max(e, 1e-99))) for e in err))
print('best  :', ' '.join('%3.0f' % s for s in best))
print('sum cold %.0f, sum best %.0f' % (sum(cold), sum(best)))
