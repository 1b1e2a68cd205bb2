"""Rate of the resident-basis evaluation (vi_eval_resident_f64) on a 256^3 grid at the default order, by timesteps per call:
K2r (csrc/vi_eval_resident.hip) and, with VINTERP_EVAL_RESIDENT=blas, the library's product.  python tools/perf_eval_resident.py [n]"""
import io
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import _lib, synth                                  # noqa: E402
from volumetricinterp_amd.models.sphharmlag import Model                      # noqa: E402

CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = Model(io.StringIO(CFG))
ctx, h, N = m.ctx, m.handle(), m.nbasis
g = synth.query_grid(n)
Q = g[0].size
dq = [ctx.to_device(a.ravel()) for a in g]
dY = ctx.empty((N, Q))
ctx.timer_start()
_lib.check(_lib.lib.vi_eval_basis_f64(h, Q, dq[0].ptr, dq[1].ptr, dq[2].ptr, None, 0, 0., dY.ptr), 'vi_eval_basis_f64')
print('basis of %d^3 points (%.1f GB): %.1f ms' % (n, N * Q * 8 / 1e9, ctx.timer_stop_ms()))
rng = np.random.default_rng(0)
Tmax = 512
dC = ctx.to_device(rng.standard_normal((Tmax, N)))
dO = ctx.empty((Tmax, Q))
for T in (64, 128, 256, 512):
    best = 1e9
    for rep in range(3):
        ctx.timer_start()
        _lib.check(_lib.lib.vi_eval_resident_f64(h, Q, T, dY.ptr, dC.ptr, dO.ptr), 'vi_eval_resident_f64')
        best = min(best, ctx.timer_stop_ms())
    print('%s T %4d: %8.3f ms  %.1f TFLOP/s  %.3e point-timesteps/s; algorithmic bytes (Y once + out) %.1f GB -> %.0f GB/s' % (
        os.environ.get('VINTERP_EVAL_RESIDENT', 'own'), T, best, 2. * N * Q * T / best / 1e9, Q * T / (best * 1e-3),
        (N * Q * 8 + T * Q * 8) / 1e9, (N * Q * 8 + T * Q * 8) / 1e9 / (best * 1e-3)))
