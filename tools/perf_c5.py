#!/usr/bin/env python3
"""BASELINE configs[4] (one GPU's share): MAXK = 8, MAXL = 12 (N = 1152; CAP_LIM = 15 so that no basis column
overflows), 64-beam x 200-range geometry, T records: fit (chi2 search, covariance) + evaluation of a
128^3 grid.  Regularisation matrix: R = I * mean|diag(A^T W A)| (synthetic, SURVEY 8d: no fixture at this order)."""
import io
import os
import sys
import time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth, _lib
from volumetricinterp_amd.fitengine import FitEngine
from volumetricinterp_amd.models.sphharmlag import Model

CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 8\nMAXL = 12\nCAP_LIM = 15\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    m = Model(io.StringIO(CFG))
    ctx = m.ctx
    ctx.eval_timing(True)                          # vi_eval_kernel_ms below
    lat, lon, alt = synth.beams(64, 200, seed=0)
    P, N = lat.size, m.nbasis
    d = [ctx.to_device(a) for a in (lat, lon, alt)]
    t0 = time.perf_counter()
    At = m.basis_device(d[0], d[1], d[2], P, transposed=True)
    ctx.sync()
    t_basis = time.perf_counter() - t0
    A = At.download().T
    assert np.all(np.isfinite(A)), 'basis has non-finite columns'
    value, error = synth.synth_records(A, T, seed0=1000)
    W = error**-2.
    R = np.eye(N) * np.mean(np.abs(np.einsum('pn,p,pn->n', A, W[0], A)))
    eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
    eng.upload_records(W, value)
    npts = [P] * T
    res = eng.fit_resident(npts, calccov=True)
    eng.stats = dict(solves=0, launches=0)
    t0 = time.perf_counter()
    res = eng.fit_resident(npts, calccov=True)
    ctx.sync()
    t_fit = time.perf_counter() - t0
    oc = res['search']['curvature']['outcomes']
    g = synth.query_grid(n)
    Q = g[0].size
    dq = [ctx.to_device(a.ravel()) for a in g]
    Cf = np.nan_to_num(res['Coeffs'])
    dC = ctx.to_device(Cf)
    dout = ctx.empty((T, Q))
    best = 1e9
    for _ in range(3):
        _lib.check(_lib.lib.vi_eval_f64(m.handle(), Q, dq[0].ptr, dq[1].ptr, dq[2].ptr, T, dC.ptr, None, 0, 0., dout.ptr), 'eval')
        ms = C.c_double()
        _lib.check(_lib.lib.vi_eval_kernel_ms(ctx.handle, C.byref(ms)), 'ms')
        best = min(best, ms.value)
    print('configs[4] share: N=%d, P=%d, T=%d: basis %.1f ms; fit %.1f ms (%d solves, %d launches, outcomes %s); '
          'eval %d^3 x %d: %.2f ms = %.3e point-timesteps/s'
          % (N, P, T, t_basis * 1e3, t_fit * 1e3, eng.stats['solves'], eng.stats['launches'],
             {o: oc.count(o) for o in set(oc)}, n, T, best, Q * T / (best * 1e-3)))


if __name__ == '__main__':
    main()
