#!/usr/bin/env python3
"""Throughput probe of the many-timesteps evaluation: T records on one n^3 grid, matrix-core kernel
vs the VALU tile kernels (VINTERP_EVAL_MFMA=0)."""
import io, os, sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth, _lib
from volumetricinterp_amd.models.sphharmlag import Model
from tools.perf_fit import CFG


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    m = Model(io.StringIO(CFG % (4, 6)))
    ctx = m.ctx
    ctx.eval_timing(True)                          # vi_eval_kernel_ms below
    h = m.handle()
    g = synth.query_grid(n)
    Q = g[0].size
    dq = [ctx.to_device(a.ravel()) for a in g]
    Cf = np.random.default_rng(0).standard_normal((T, m.nbasis))
    dC = ctx.to_device(Cf)
    dout = ctx.empty((T, Q))
    best = 1e9
    for rep in range(4):
        _lib.check(_lib.lib.vi_eval_f64(h, Q, dq[0].ptr, dq[1].ptr, dq[2].ptr, T, dC.ptr, None, 0, 0., dout.ptr), 'eval')
        ms = C.c_double()
        _lib.check(_lib.lib.vi_eval_kernel_ms(ctx.handle, C.byref(ms)), 'ms')
        best = min(best, ms.value)
    print('T=%d grid %d^3: %.3f ms -> %.3e point-timesteps/s (%.1f %% of the 78.6 TF fp64 peak at 2N flop/pair; MFMA=%s)'
          % (T, n, best, Q * T / (best * 1e-3), 100 * Q * T * 2 * m.nbasis / (best * 1e-3) / 78.6e12,
             os.environ.get('VINTERP_EVAL_MFMA', '1')))


if __name__ == '__main__':
    main()
