#!/usr/bin/env python3
"""BASELINE configs[3], one GPU's share: T timesteps (10000 / 8 = 1250) evaluated on a 256^3 geodetic grid.
The output of one timestep tile stays resident; the tile buffer is reused (1250 x 256^3 x 8 B = 168 GB would fit the
288 GB of an MI355X, but nothing downstream of this probe reads it)."""
import ctypes as C
import io
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth, _lib
from volumetricinterp_amd.models.sphharmlag import Model
from tools.perf_fit import CFG


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    m = Model(io.StringIO(CFG % (4, 6)))
    ctx = m.ctx
    ctx.eval_timing(True)                          # vi_eval_kernel_ms below
    h = m.handle()
    g = synth.query_grid(n)
    Q = g[0].size
    dq = [ctx.to_device(a.ravel()) for a in g]
    Cf = np.random.default_rng(0).standard_normal((T, m.nbasis))
    dC = ctx.to_device(Cf)
    dout = ctx.empty((chunk, Q))
    ctx.sync()
    kern = 0.
    t0 = time.perf_counter()
    for s in range(0, T, chunk):
        tc = min(chunk, T - s)
        _lib.check(_lib.lib.vi_eval_f64(h, Q, dq[0].ptr, dq[1].ptr, dq[2].ptr, tc, dC.offset_ptr(s * m.nbasis), None, 0, 0.,
                                        dout.ptr), 'eval')
        ms = C.c_double()
        _lib.check(_lib.lib.vi_eval_kernel_ms(ctx.handle, C.byref(ms)), 'ms')
        kern += ms.value
    ctx.sync()
    wall = time.perf_counter() - t0
    print('configs[3] share: T=%d timesteps x %d^3 points: kernels %.1f ms, wall %.1f ms -> %.3e point-timesteps/s '
          '(%.1f %% of 78.6 TF at 2N flop per pair); output %.1f GB per chunk of %d'
          % (T, n, kern, wall * 1e3, Q * T / wall, 100 * Q * T * 2 * m.nbasis / wall / 78.6e12, chunk * Q * 8 / 1e9, chunk))


if __name__ == '__main__':
    main()
