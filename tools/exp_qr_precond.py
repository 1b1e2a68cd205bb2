"""Experiment (round 3): precondition the cold truncated solve by ONE pivoted-QR similarity step.
X P = Q R (Householder QR with column pivoting), X1 = Q^T X Q (same eigenvalues), y1 = Q^T y, solve X1 c1 = y1 with the
in-LDS Jacobi kernel, c = Q c1.  Here Q comes from SciPy on the host and the products from NumPy: the point is the sweep
count of the GPU kernel on X1 and the accuracy of the whole against 50-digit arithmetic (tests/golden/exact_default_c2.npz).
    python tools/exp_qr_precond.py"""
import os
import sys

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                                   # noqa: E402
from volumetricinterp_amd import _lib, fitengine       # noqa: E402,F401  (fitengine declares the argument types)

EPS = float(np.finfo(float).eps)
e = np.load(os.path.join(ROOT, 'tests', 'golden', 'exact_default_c2.npz'))
f = np.load(os.path.join(ROOT, 'tests', 'golden', 'fit_default_c2.npz'), allow_pickle=True)
A = oracle.SphHarmLagOracle().basis(f['lat'], f['lon'], f['alt'])
ctx = _lib.get_context()
rel = lambda x, y: float(np.linalg.norm(x - y) / np.linalg.norm(y))      # noqa: E731


def gpu_solve(X, y):
    B, N = X.shape[0], X.shape[1]
    dX, dy = ctx.to_device(X.copy()), ctx.to_device(y)
    dC, drank = ctx.empty((B, N)), ctx.empty((B,), np.int32)
    _lib.check(_lib.lib.vi_solve_trunc_f64(ctx.handle, B, N, dX.ptr, dy.ptr, None, EPS, dC.ptr, drank.ptr, N * EPS, None))
    dX2, dl, ds = ctx.to_device(X.copy()), ctx.empty((B, N)), ctx.empty((B,), np.int32)
    _lib.check(_lib.lib.vi_eigvals_f64(ctx.handle, B, N, dX2.ptr, dl.ptr, ds.ptr))
    return dC.download(), drank.download(), ds.download()


X, y = e['X'], e['y']
B, N = X.shape[0], X.shape[1]
X1, y1, Qs = np.empty_like(X), np.empty_like(y), []
for i in range(B):
    mx = np.max(np.abs(X[i]))
    sc = 2.0**(1 - np.frexp(mx)[1])
    Q, R, P = sl.qr(X[i] * sc, pivoting=True)
    M = (Q.T @ (X[i] * sc) @ Q) / sc
    X1[i] = 0.5 * (M + M.T)
    y1[i] = Q.T @ y[i]
    Qs.append(Q)
for tag, (C, rank, sw) in (('plain', gpu_solve(X, y)), ('QR-preconditioned', gpu_solve(X1, y1))):
    for i in range(B):
        Ci = C[i] if tag == 'plain' else Qs[i] @ C[i]
        t = int(e['record'][i])
        b, W = f['value'][t], f['error'][t]**-2.
        chi = float(sum((A @ Ci - b)**2 * W))
        print('%-18s sys %d log10 alpha %.3f: sweeps %2d rank %d (exact %d) chi2 rel err %.1e rel(A c) %.1e rel(c) %.1e'
              % (tag, i, e['log10_alpha'][i], sw[i], rank[i], e['rank'][i], abs(chi - e['chi2'][i]) / e['chi2'][i],
                 rel(A @ Ci, A @ e['C'][i]), rel(Ci, e['C'][i])))
# the walk systems of the bench record: sweeps decade by decade
AWA, R, yv = f['rec0_AWA'], f['R'], f['rec0_y']
las = np.arange(0, -60, -2)
Xw = np.array([AWA + 10.0**la * R for la in las])
Xp = np.empty_like(Xw)
for i in range(len(las)):
    sc = 2.0**(1 - np.frexp(np.max(np.abs(Xw[i])))[1])
    Q, _, _ = sl.qr(Xw[i] * sc, pivoting=True)
    M = (Q.T @ (Xw[i] * sc) @ Q) / sc
    Xp[i] = 0.5 * (M + M.T)
yw = np.tile(yv, (len(las), 1))
_, r0, s0 = gpu_solve(Xw, yw)
_, r1, s1 = gpu_solve(Xp, yw)
print('walk decades      ', las.tolist())
print('sweeps plain      ', s0.tolist(), 'sum', int(s0.sum()))
print('sweeps QR-precond ', s1.tolist(), 'sum', int(s1.sum()))
print('rank plain        ', r0.tolist())
print('rank QR-precond   ', r1.tolist())
