#!/usr/bin/env python3
"""Where the hull-mask pass spends its time: the call with F facets minus the call without a hull, for facet lists that
switch phases of k_hull_mask on and off (few facets: geodetic -> ECEF only; every point deep inside: the whole facet loop,
no early exit, no fp64 recheck; every point outside: early exit at once)."""
import io, os, sys
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
eq, tol = hull_equations(R[ConvexHull(R).vertices])
g = synth.query_grid(n); Q = g[0].size
d = [ctx.to_device(a.ravel()) for a in g]
C = ctx.to_device(np.random.default_rng(0).standard_normal((1, 144)))
out = ctx.empty((1, Q))


def call_ms(e):
    de = ctx.to_device(np.ascontiguousarray(e)) if e is not None else None

    def run():
        _lib.check(_lib.lib.vi_eval_f64(h, Q, d[0].ptr, d[1].ptr, d[2].ptr, 1, C.ptr, de.ptr if de is not None else None,
                                        0 if e is None else len(e), tol, out.ptr), 'eval')
    run(); ctx.sync(); ctx.timer_start()
    for _ in range(10): run()
    ms = ctx.timer_stop_ms() / 10
    return ms, float(np.isfinite(out.download()).mean())


base, _ = call_ms(None)
inside = eq.copy(); inside[:, 3] -= 2e6       # (2000 km: the fp16 planes of k_hull_mask_mx reach 16 000 km from c0)
outside = eq.copy(); outside[:, 3] += 2e6
for name, e in (('460 facets as they are', eq), ('16 facets', eq[:16]), ('460 facets, all points deep inside', inside),
                ('460 facets, all points outside', outside), ('920 facets, all inside', np.vstack([inside, inside]))):
    ms, frac = call_ms(e)
    print('%-40s call %.3f ms, over the call without hull (%.3f): %.1f us; inside fraction %.3f' %
          (name, ms, base, (ms - base) * 1e3, frac))
