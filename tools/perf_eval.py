#!/usr/bin/env python3
"""Quick throughput probe of the fused evaluation kernel (device-resident inputs)."""
import io
import sys
import os
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth, _lib
from volumetricinterp_amd.models.sphharmlag import Model

CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = %d\nMAXL = %d\nCAP_LIM = %g\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    k, l, cap = (int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])) if len(sys.argv) > 5 else (4, 6, 10.)
    m = Model(io.StringIO(CFG % (k, l, cap)))
    h = m.handle()
    ctx = m.ctx
    g = synth.query_grid(n)
    Q = g[0].size
    d = [ctx.to_device(a.ravel()) for a in g]
    rng = np.random.default_rng(0)
    C = ctx.to_device(rng.standard_normal((T, m.nbasis)))
    out = ctx.empty((T, Q))
    def run():
        _lib.check(_lib.lib.vi_eval_f64(h, Q, d[0].ptr, d[1].ptr, d[2].ptr, T, C.ptr, None, 0, 0., out.ptr), 'eval')
    for _ in range(3):
        run()
    ctx.sync()
    reps = 10
    ctx.timer_start()
    for _ in range(reps):
        run()
    ms = ctx.timer_stop_ms() / reps
    print('eval n=%d Q=%d T=%d order=(%d,%d,cap %g): %.3f ms  -> %.3e point-timesteps/s, %.1f GB/s algorithmic'
          % (n, Q, T, k, l, cap, ms, Q * T / ms * 1e3, Q * (24 + 8 * T) / ms * 1e3 / 1e9))


if __name__ == '__main__':
    main()
