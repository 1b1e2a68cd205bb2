"""Diagnostic: Jacobi sweeps / time on the systems the default-order search actually solves."""
import os, sys, numpy as np, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden
from volumetricinterp_amd import _lib, fitengine
ctx = _lib.get_context()
f = load_golden('fit_default'); R = f['R']
# rebuild AWA of record 0 from the fixture (rec0_AWA)
AWA = f['rec0_AWA']; N = AWA.shape[0]
al = np.arange(0, -60, -1.0)
X = np.stack([AWA + 10.0**a * R for a in al])
dX = ctx.to_device(X); dl = ctx.empty((len(al), N)); ds = ctx.empty((len(al),), np.int32)
_lib.check(_lib.lib.vi_eigvals_f64(ctx.handle, len(al), N, dX.ptr, dl.ptr, ds.ptr), 'eig')
sw = ds.download(); lam = dl.download()
print('sweeps per alpha:', sw.tolist())
for i in (0, 20, 29, 40, 59):
    ref = np.linalg.eigvalsh(X[i]); mine = np.sort(lam[i])
    print('alpha 1e%d: max |dlam|/max|lam| = %.2e' % (al[i], np.max(np.abs(mine - ref)) / np.max(np.abs(ref))))
for B in (1, 60):
    dX.upload(X)
    ctx.timer_start()
    _lib.check(_lib.lib.vi_eigvals_f64(ctx.handle, B, N, dX.ptr, dl.ptr, ds.ptr), 'eig')
    print('B=%d: %.2f ms' % (B, ctx.timer_stop_ms()))
