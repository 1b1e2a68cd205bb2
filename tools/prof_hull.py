#!/usr/bin/env python3
"""One evaluation call with the hull mask, 50 times - for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_hull.py`:
the kernels of the call (k_prep_hull, the mask pass, k_prep_coef, the evaluation kernel) with their durations."""
import io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth, _lib
from volumetricinterp_amd.models.sphharmlag import Model
from volumetricinterp_amd.estimate import hull_equations, order_facets
from volumetricinterp_amd.geodesy import geodetic2ecef
from scipy.spatial import ConvexHull
CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = Model(io.StringIO(CFG)); h = m.handle(); ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
R = np.array(geodetic2ecef(lat, lon, alt)).T
hv = R[ConvexHull(R).vertices]
eq, tol = hull_equations(hv)
eq = order_facets(eq, hv)
g = synth.query_grid(n); Q = g[0].size
d = [ctx.to_device(a.ravel()) for a in g]
C = ctx.to_device(np.random.default_rng(0).standard_normal((1, 144)))
out = ctx.empty((1, Q))
de = ctx.to_device(np.ascontiguousarray(eq))


def run():
    _lib.check(_lib.lib.vi_eval_f64(h, Q, d[0].ptr, d[1].ptr, d[2].ptr, 1, C.ptr, de.ptr, len(eq), tol, out.ptr), 'eval')


run(); ctx.sync(); ctx.timer_start()
for _ in range(50): run()
print('call with hull mask: %.4f ms, inside fraction %.4f' % (ctx.timer_stop_ms() / 50, float(np.isfinite(out.download()).mean())))
