"""Diagnostic (GPU): chi^2 of the cold truncated solve at given log10(alpha) for one record of a default-order fixture,
beside the same quantity computed on the host (LAPACK eigh / lstsq) from the GPU's own normal equations and basis, so
that solver error and formation error can be told apart.  Usage: python tools/diag_chi2_point.py fixture rec la1 la2 ..."""
import os
import sys
import tempfile

import numpy as np
import scipy.linalg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden                            # noqa: E402
from test_gpu_fit import make_interp                         # noqa: E402
from volumetricinterp_amd import _lib                        # noqa: E402
from volumetricinterp_amd.fitengine import FitEngine         # noqa: E402
import ctypes as C                                           # noqa: E402

name, t = sys.argv[1], int(sys.argv[2])
las = [float(x) for x in sys.argv[3:]]
f = load_golden(name)
it = make_interp(tempfile.mkdtemp(), str(f['cfg']))
ctx = it.ctx
lat, lon, alt = f['lat'], f['lon'], f['alt']
P = lat.size
it.model.handle(ctx)
At = it.model.basis_device(ctx.to_device(lat), ctx.to_device(lon), ctx.to_device(alt), P, transposed=True)
A = At.download().T
W = f['error']**-2.
b = f['value']
R = f['R']
eng = FitEngine(ctx, At, P, it.model.nbasis, {'curvature': R}, ['curvature'])
eng.load_records(W, b)
AWA, y = eng.normal_equations()
AWAh = np.einsum('ji,j,jk->ik', A, W[t], A)
yh = np.einsum('ji,j,j->i', A, W[t], b[t])
print('max sweeps env', os.environ.get('VINTERP_MAX_SWEEPS'), '| rel(AWA gpu vs einsum on the gpu basis) %.2e  rel(y) %.2e'
      % (np.linalg.norm(AWA[t] - AWAh) / np.linalg.norm(AWAh), np.linalg.norm(y[t] - yh) / np.linalg.norm(yh)))
eps = np.finfo(float).eps


def chi2(Cv):
    return float(sum((A @ Cv - b[t])**2 * W[t]))


gpu = eng.chi2_batch(np.full(len(las), t, dtype=np.int32), {'curvature': np.power(10., np.array(las))})
N = it.model.nbasis
for la, g in zip(las, gpu):
    X = AWA[t] + 10.**la * R
    lam, V = np.linalg.eigh(X)
    keep = np.abs(lam) > eps * np.abs(lam).max()
    c_eigh = chi2((V[:, keep] / lam[keep]) @ (V[:, keep].T @ y[t]))
    c_lstsq = chi2(scipy.linalg.lstsq(X, y[t])[0])
    # the solver's own view of the system: eigenvalues + sweeps
    dX = ctx.to_device(X[None].copy())
    dl, ds = ctx.empty((1, N)), ctx.empty((1,), np.int32)
    _lib.check(_lib.lib.vi_eigvals_f64(ctx.handle, 1, N, dX.ptr, dl.ptr, ds.ptr), 'vi_eigvals_f64')
    lg = np.sort(np.abs(dl.download()[0]))[::-1]
    le = np.sort(np.abs(lam))[::-1]
    k = int(keep.sum())
    print('log10a %.5f: gpu chi2 %.4f | host on the gpu AWA: eigh %.4f (rank %d)  lstsq %.4f | jacobi sweeps %d, rank %d; '
          'eigenvalues near the cut gpu %s | eigh %s'
          % (la, g, c_eigh, k, c_lstsq, int(ds.download()[0]), int((lg > eps * lg[0]).sum()),
             ' '.join('%.3e' % (x / lg[0]) for x in lg[k - 3:k + 2]), ' '.join('%.3e' % (x / le[0]) for x in le[k - 3:k + 2])))
