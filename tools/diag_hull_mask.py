#!/usr/bin/env python3
"""The hull mask of the device against the fp64 definition, point by point: where they differ, which facet holds the true
maximum (tile and row of the 32-facet operand tiles of k_hull_mask_mx) and where the point sits in its workgroup (lane, set)."""
import io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth, _lib
from volumetricinterp_amd.models.sphharmlag import Model
from volumetricinterp_amd.estimate import hull_equations, order_facets
from volumetricinterp_amd.geodesy import geodetic2ecef
from scipy.spatial import ConvexHull
CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = Model(io.StringIO(CFG)); h = m.handle(); ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
R = np.array(geodetic2ecef(lat, lon, alt)).T
hv = R[ConvexHull(R).vertices]
eq, tol = hull_equations(hv)
eq = order_facets(eq, hv)
g = synth.query_grid(n); Q = g[0].size
X = np.array(geodetic2ecef(*[a.ravel() for a in g])).T
d_all = X @ eq[:, :3].T + eq[:, 3]
d = d_all.max(axis=1); fstar = d_all.argmax(axis=1)
expect = d <= tol
dev = [ctx.to_device(a.ravel()) for a in g]
C = ctx.to_device(np.zeros((1, 144)))
out = ctx.empty((1, Q))
for F in (len(eq), 64, 32, 16):
    e = np.ascontiguousarray(eq[:F])
    dd = d_all[:, :F]; dm = dd.max(axis=1); fs = dd.argmax(axis=1); ex = dm <= tol
    de = ctx.to_device(e)
    _lib.check(_lib.lib.vi_eval_f64(h, Q, dev[0].ptr, dev[1].ptr, dev[2].ptr, 1, C.ptr, de.ptr, F, tol, out.ptr), 'eval')
    got = np.isfinite(out.download()[0])
    bad = np.nonzero(got != ex)[0]
    print('F = %d: %d of %d points differ (device inside, truly outside: %d; device outside, truly inside: %d)' %
          (F, len(bad), Q, int((got & ~ex).sum()), int((~got & ex).sum())))
    if len(bad):
        print('   |d - tol| of the differing points: min %.3g median %.3g max %.3g m' %
              (np.abs(dm[bad] - tol).min(), np.median(np.abs(dm[bad] - tol)), np.abs(dm[bad] - tol).max()))
        print('   tile of the deciding facet:', np.bincount(fs[bad] // 32, minlength=(F + 31) // 32))
        print('   row (facet mod 32) of the deciding facet:', np.bincount(fs[bad] % 32, minlength=32))
        print('   lane half of the point (q mod 64 >= 32):', np.bincount((bad % 64) // 32, minlength=2),
              ' set u (q mod 2048 // 256):', np.bincount((bad % 2048) // 256, minlength=8))
        print('   first differing points:', bad[:12], 'deciding facets', fs[bad[:12]])
