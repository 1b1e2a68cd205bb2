"""Throughput of the radial-basis evaluation kernel k_eval_rbf (radbasfun.py:83-112 + estimate.py:110-123 fused):
out[q] = sum_n C[n] exp(-|x_q - c_n|^2 / eps^2) over a 128^3 grid.  Usage (GPU): python tools/perf_rbf.py [ngrid] [n]"""
import io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import _lib, synth
from volumetricinterp_amd.models.radbasfun import Model
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
CFG = ('[MODEL]\nNAME = radbasfun\nLATCP = 78\nLONCP = 262\nEPS = 2.5e5\nLATRANGE = 72, 84\nLONRANGE = 240, 284\n'
       'ALTRANGE = 100, 700\nNUMGRIDPNT = %d\n' % ng)
m = Model(io.StringIO(CFG))
ctx = m.ctx
N = m.nbasis
g = np.linspace(0., 1., n)
lat, lon, alt = np.meshgrid(74. + 8. * g, 245. + 30. * g, 1e5 + 5e5 * g, indexing='ij')
Q = lat.size
d = [ctx.to_device(a.ravel()) for a in (lat, lon, alt)]
rng = np.random.default_rng(0)
for T in (1, 4, 16):
    C = rng.standard_normal((T, N))
    dC = ctx.to_device(C)
    out = ctx.empty((T, Q))
    best = 1e9
    for rep in range(4):
        ctx.timer_start()
        _lib.check(_lib.lib.vi_eval_f64(m.handle(), Q, d[0].ptr, d[1].ptr, d[2].ptr, T, dC.ptr, None, 0, 0., out.ptr), 'eval')
        best = min(best, ctx.timer_stop_ms())
    print('radbasfun N=%d, %d^3 points, T=%d: %.3f ms = %.3e point-timesteps/s, %.3e (point, centre) pairs/s'
          % (N, n, T, best, Q * T / best * 1e3, Q * N * ((T + 3) // 4) / best * 1e3))
