"""Diagnosis: default16 records the engine flags as a jump - GPU eigenvalues / rank / chi^2 of X(alpha) at the returned alpha
and 1e-6 decades either side, against the accurate CPU evaluation (oracle/accurate.py).  python tools/diag_jump.py"""
import math
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                                     # noqa: E402
from oracle import accurate                                       # noqa: E402
from volumetricinterp_amd import _lib, fitengine                 # noqa: E402,F401
from volumetricinterp_amd.interpolate import Interpolate          # noqa: E402

EPS = float(np.finfo(float).eps)
f = np.load(os.path.join(ROOT, 'tests', 'golden', 'fit_default16.npz'), allow_pickle=True)
with tempfile.TemporaryDirectory() as td:
    cfg = os.path.join(td, 'c.ini')
    open(cfg, 'w').write(str(f['cfg']))
    it = Interpolate(cfg)
    res = it.fit_records(f['lat'], f['lon'], f['alt'], f['value'], f['error'], {'curvature': f['R']})
A = oracle.SphHarmLagOracle().basis(f['lat'], f['lon'], f['alt'])
ctx = _lib.get_context()
info = res['search']['curvature']['info']
N = 144
for t in range(16):
    if not info[t].get('jump'):
        continue
    a = res['reg_params'][t]['curvature']
    la = math.log10(a)
    b, W = f['value'][t], f['error'][t]**-2.
    AWA = np.einsum('ji,j,jk->ik', A, W, A)
    y = np.einsum('ji,j,j->i', A, W, b)
    print('record %d: log10 alpha %.12f, GPU final chi2 %.4f, nu %.1f, other_end %r' % (t, la, res['chi_sq'][t], info[t]['sf'] * 550,
                                                                                      info[t].get('other_end')))
    for dl in (-1e-6, 0.0, 1e-6):
        X = AWA + 10.**(la + dl) * f['R']
        dX, dy = ctx.to_device(X[None].copy()), ctx.to_device(y[None])
        dC, drk = ctx.empty((1, N)), ctx.empty((1,), np.int32)
        _lib.check(_lib.lib.vi_solve_trunc_f64(ctx.handle, 1, N, dX.ptr, dy.ptr, None, EPS, dC.ptr, drk.ptr, N * EPS, None))
        Cg = dC.download()[0]
        dX2, dl_, ds = ctx.to_device(X[None].copy()), ctx.empty((1, N)), ctx.empty((1,), np.int32)
        _lib.check(_lib.lib.vi_eigvals_f64(ctx.handle, 1, N, dX2.ptr, dl_.ptr, ds.ptr))
        lg = np.sort(np.abs(dl_.download()[0]))[::-1]
        ca, rka = accurate.lstsq_accurate(X, y)
        sc = 2.0**(1 - np.frexp(np.max(np.abs(X)))[1])
        Q, _, _ = __import__('scipy.linalg').linalg.qr(X * sc, pivoting=True)
        X1 = Q.T @ (X * sc) @ Q
        lc, _, _ = accurate.jacobi_eigh(0.5 * (X1 + X1.T))
        lc = np.sort(np.abs(lc))[::-1] / sc
        chi = lambda c: float(np.sum((A @ c - b)**2 * W))      # noqa: E731
        k = rka
        print('  dl %+.0e: GPU rank %d sweeps %d chi2 %.4f | accurate rank %d chi2 %.4f | |lam|/max/eps around the cut: GPU %s  CPU %s'
              % (dl, drk.download()[0], ds.download()[0], chi(Cg), rka, chi(ca),
                 np.round(lg[k - 3:k + 3] / lg[0] / EPS, 4), np.round(lc[k - 3:k + 3] / lc[0] / EPS, 4)))
