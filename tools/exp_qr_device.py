"""K3p on the device (csrc/vi_qr.hip): stage check of vi_qr_similarity_f64, then the cold truncated solve with and without
the pre-conditioner (VINTERP_QRPRE=0 / 1, one process each) on the default-order systems of tests/golden: accuracy against
50-digit arithmetic, sweeps, time per launch for 1 / 256 / 1024 systems.
    python tools/exp_qr_device.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                                   # noqa: E402
from volumetricinterp_amd import _lib, fitengine               # noqa: E402,F401
import ctypes as C                                              # noqa: E402

EPS = float(np.finfo(float).eps)
e = np.load(os.path.join(ROOT, 'tests', 'golden', 'exact_default_c2.npz'))
f = np.load(os.path.join(ROOT, 'tests', 'golden', 'fit_default_c2.npz'), allow_pickle=True)
A = oracle.SphHarmLagOracle().basis(f['lat'], f['lon'], f['alt'])
ctx = _lib.get_context()
rel = lambda x, y: float(np.linalg.norm(x - y) / np.linalg.norm(y))      # noqa: E731
print('VINTERP_QRPRE =', os.environ.get('VINTERP_QRPRE', '(unset: on)'))

X, y = e['X'], e['y']
B, N = X.shape[0], X.shape[1]
if os.environ.get('VINTERP_QRPRE', '1') != '0':
    Xs = np.array([x * 2.0**(1 - np.frexp(np.max(np.abs(x)))[1]) for x in X])
    dX, dy = ctx.to_device(Xs), ctx.to_device(y)
    dX1, dy1, dQ = ctx.empty((B, N, N)), ctx.empty((B, N)), ctx.empty((B, N, N))
    _lib.check(_lib.lib.vi_qr_similarity_f64(ctx.handle, B, N, dX.ptr, dy.ptr, dX1.ptr, dy1.ptr, dQ.ptr))
    X1, y1, Q = dX1.download(), dy1.download(), dQ.download()
    for i in range(B):
        Qi = Q[i].T                                   # Q[:, j] contiguous -> rows of the download
        Xl = np.tril(Xs[i]) + np.tril(Xs[i], -1).T
        ref = Qi.T @ Xl @ Qi
        print('sys %d: |Q^T Q - I| %.1e  |X1 - Q^T X Q|/|X| %.1e  |y1 - Q^T y|/|y| %.1e  asym %.1e  eig diff %.1e'
              % (i, np.max(np.abs(Qi.T @ Qi - np.eye(N))), np.max(np.abs(X1[i] - ref)) / np.max(np.abs(Xl)),
                 rel(y1[i], Qi.T @ y[i]), np.max(np.abs(X1[i] - X1[i].T)),
                 np.max(np.abs(np.sort(np.linalg.eigvalsh(np.tril(X1[i]) + np.tril(X1[i], -1).T)) - np.sort(np.linalg.eigvalsh(Xl))))))


def gpu_solve(X, y, want_H=False):
    B, N = X.shape[0], X.shape[1]
    dX, dy = ctx.to_device(X.copy()), ctx.to_device(y)
    dC, drank = ctx.empty((B, N)), ctx.empty((B,), np.int32)
    dH = ctx.empty((B, N, N)) if want_H else None
    _lib.check(_lib.lib.vi_solve_trunc_f64(ctx.handle, B, N, dX.ptr, dy.ptr, None, EPS, dC.ptr, drank.ptr, N * EPS,
                                           dH.ptr if want_H else None))
    dX2, dl, ds = ctx.to_device(X.copy()), ctx.empty((B, N)), ctx.empty((B,), np.int32)
    _lib.check(_lib.lib.vi_eigvals_f64(ctx.handle, B, N, dX2.ptr, dl.ptr, ds.ptr))
    return dC.download(), drank.download(), ds.download(), (dH.download() if want_H else None)


for want_H in (False, True):
    Cg, rank, sw, H = gpu_solve(X, y, want_H)
    for i in range(B):
        t = int(e['record'][i])
        b, W = f['value'][t], f['error'][t]**-2.
        chi = float(sum((A @ Cg[i] - b)**2 * W))
        extra = ''
        if want_H:
            Hl = np.linalg.pinv(X[i], rcond=N * EPS)
            extra = '  H sym %.1e  X H X - X %.1e' % (np.max(np.abs(H[i] - H[i].T)) / np.max(np.abs(H[i])),
                                                      np.max(np.abs(X[i] @ H[i] @ X[i] - X[i])) / np.max(np.abs(X[i])))
        print('%s sys %d log10 alpha %.3f: sweeps %2d rank %d (exact %d) chi2 rel err %.1e rel(A c) %.1e rel(c) %.1e%s'
              % ('with H' if want_H else 'C only', i, e['log10_alpha'][i], sw[i], rank[i], e['rank'][i],
                 abs(chi - e['chi2'][i]) / e['chi2'][i], rel(A @ Cg[i], A @ e['C'][i]), rel(Cg[i], e['C'][i]), extra))
# timing: the walk systems of the bench record
AWA, R, yv = f['rec0_AWA'], f['R'], f['rec0_y']
for nB in (1, 32, 256, 1024):
    las = -np.linspace(20, 60, nB) if nB > 1 else np.array([-26.3])
    Xw = np.array([AWA + 10.0**la * R for la in las])
    yw = np.tile(yv, (nB, 1))
    dX0 = ctx.to_device(Xw)
    dXw, dyw = ctx.empty((nB, N, N)), ctx.to_device(yw)
    dC, drank = ctx.empty((nB, N)), ctx.empty((nB,), np.int32)
    best = 1e9
    for rep in range(4):
        _lib.check(_lib.lib.vi_d2h(ctx.handle, Xw.ctypes.data_as(_lib.VOIDP), dX0.ptr, 8))     # sync
        dXw.upload(Xw)
        _lib.check(_lib.lib.vi_timer_start(ctx.handle))
        _lib.check(_lib.lib.vi_solve_trunc_f64(ctx.handle, nB, N, dXw.ptr, dyw.ptr, None, EPS, dC.ptr, drank.ptr, N * EPS, None))
        ms = C.c_double()
        _lib.check(_lib.lib.vi_timer_stop_ms(ctx.handle, C.byref(ms)))
        best = min(best, ms.value)
    print('cold solve of %4d systems: %.3f ms = %.2f us per system' % (nB, best, best * 1e3 / nB))
