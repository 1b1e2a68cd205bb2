"""Diagnostic: do the walk systems X_j(10^k) = A^T W_j A + 10^k R of DIFFERENT records converge fast from the eigenbasis
of a reference system of the same decade (mean weights over a few records: geometry-level information only)?
Prints Jacobi sweeps cold / from the shared basis and the relative chi^2 difference of the two solves.
Usage (GPU): python tools/exp_shared_basis.py"""
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
T = 6
value, error = synth.synth_records(A, T, seed0=1000)
W = error**-2.
AWA = np.stack([(A.T * W[j]) @ A for j in range(T)]); y = np.stack([A.T @ (W[j] * value[j]) for j in range(T)])
ref = np.mean(AWA[3:], axis=0)                      # the shared basis comes from records 3..5, the test records are 0..2
EPS = np.finfo(float).eps
ROUNDS = 2 * ((N + 3) // 4)
ctx.solve_timing(1)


def rounds():
    ctx.sync()
    return ctx.solve_timing(1)['rounds'] / ROUNDS


def chi2(C, j):
    return float(np.sum((A @ C - value[j])**2 * W[j]))


dAWA = ctx.to_device(np.concatenate([AWA, ref[None]])); dy = ctx.to_device(np.concatenate([y, y[:1]])); dR = ctx.to_device(R)
dX = ctx.empty((1, N, N)); dC = ctx.empty((1, N)); drk = ctx.empty((1,), np.int32)
dV = ctx.empty((1, N, N)); dD1 = ctx.empty((1, N, N)); dD2 = ctx.empty((1, N, N)); dyt = ctx.empty((1, N)); dC0 = ctx.empty((1, N))
slot0 = ctx.to_device(np.zeros(1, np.int32))
print('%4s | %s' % ('k', ' | '.join('rec %d: cold  shared  dchi2/chi2 ' % j for j in range(3))))
tot_c = tot_s = 0
for k in list(range(0, -50, -2)) + [-60, -80, -100]:
    al = ctx.to_device(np.array([10.0**k]))
    dref = ctx.to_device(np.array([T], np.int32))
    _lib.check(_lib.lib.vi_warm_prepare_f64(h, 1, N, dAWA.ptr, dref.ptr, al.ptr, dR.ptr, dy.ptr,
                                            EPS, dC0.ptr, drk.ptr, dV.ptr, dD1.ptr, dD2.ptr, dyt.ptr), 'prep')
    V = dV.download()[0]
    # which orientation does the library store V in?  D2 = V^T R V must be reproduced
    D2 = dD2.download()[0]
    Vm = V if np.linalg.norm(V.T @ R @ V - D2) < np.linalg.norm(V @ R @ V.T - D2) else V.T
    row = []
    for j in range(3):
        rj = ctx.to_device(np.array([j], np.int32))
        _lib.check(_lib.lib.vi_form_system_f64(h, 1, N, dAWA.ptr, rj.ptr, al.ptr, dR.ptr, dX.ptr), 'form')
        rounds()
        _lib.check(_lib.lib.vi_solve_trunc_f64(h, 1, N, dX.ptr, dy.ptr, rj.ptr, EPS, dC.ptr, drk.ptr, N * EPS, None), 'cold')
        sc = rounds(); c2c = chi2(dC.download()[0], j)
        dD1.upload((Vm.T @ AWA[j] @ Vm)[None]); dyt.upload((Vm.T @ y[j])[None])
        rounds()
        _lib.check(_lib.lib.vi_warm_solve_f64(h, 1, N, dD1.ptr, dD2.ptr, dyt.ptr, dV.ptr, slot0.ptr, al.ptr, EPS, dC.ptr, drk.ptr, None), 'warm')
        ss = rounds(); c2s = chi2(dC.download()[0], j)
        tot_c += sc; tot_s += ss
        row.append('       %4.0f  %5.0f   %9.1e ' % (sc, ss, abs(c2s - c2c) / c2c))
    print('%4d | %s' % (k, ' | '.join(row)))
print('total sweeps cold %.0f, from the shared basis %.0f' % (tot_c, tot_s))
