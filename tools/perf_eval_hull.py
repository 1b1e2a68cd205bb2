#!/usr/bin/env python3
"""Estimate.__call__-style evaluation with the convex-hull mask (the reference's default check_hull=True)."""
import io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth, _lib
from volumetricinterp_amd.models.sphharmlag import Model
from volumetricinterp_amd.estimate import hull_equations
from volumetricinterp_amd.geodesy import geodetic2ecef
from scipy.spatial import ConvexHull
CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = Model(io.StringIO(CFG)); h = m.handle(); ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
R = np.array(geodetic2ecef(lat, lon, alt)).T
hv = R[ConvexHull(R).vertices]
eq, tol = hull_equations(hv)
print('hull vertices', len(hv), 'facet equations', eq.shape)
g = synth.query_grid(n); Q = g[0].size
d = [ctx.to_device(a.ravel()) for a in g]
C = ctx.to_device(np.random.default_rng(0).standard_normal((1, 144)))
deq = ctx.to_device(eq); out = ctx.empty((1, Q))
for F, ptr in ((0, None), (eq.shape[0], deq.ptr)):
    def run():
        _lib.check(_lib.lib.vi_eval_f64(h, Q, d[0].ptr, d[1].ptr, d[2].ptr, 1, C.ptr, ptr, F, tol, out.ptr), 'eval')
    run(); ctx.sync(); ctx.timer_start()
    for _ in range(5): run()
    ms = ctx.timer_stop_ms() / 5
    o = out.download()
    print('F=%d: %.3f ms -> %.3e points/s; inside fraction %.3f' % (F, ms, Q / ms * 1e3, np.isfinite(o).mean()))
