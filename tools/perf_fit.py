#!/usr/bin/env python3
"""Throughput probe of the batched fit (BASELINE configs[2]: T records sharing one 26x100 geometry)."""
import io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth, _lib
from volumetricinterp_amd.fitengine import FitEngine
from volumetricinterp_amd.models.sphharmlag import Model

CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = %d\nMAXL = %d\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    k, l = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4, 6)
    cov = (sys.argv[4] != 'nocov') if len(sys.argv) > 4 else True
    m = Model(io.StringIO(CFG % (k, l)))
    ctx = m.ctx
    lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
    P, N = lat.size, m.nbasis
    d = [ctx.to_device(a) for a in (lat, lon, alt)]
    At = m.basis_device(d[0], d[1], d[2], P, transposed=True)
    A = At.download().T
    R = m.eval_reg_matricies['curvature']()
    value, error = synth.synth_records(A, T, seed0=1000)
    eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
    eng.upload_records(error**-2., value)
    npts = [P] * T
    t0 = time.perf_counter(); res = eng.fit_resident(npts, calccov=cov); t1 = time.perf_counter()
    eng.stats = dict(solves=0, launches=0)
    t0 = time.perf_counter(); res = eng.fit_resident(npts, calccov=cov); ctx.sync(); t1 = time.perf_counter()
    oc = res['search']['curvature']['outcomes']
    print('T=%d order (%d,%d) N=%d cov=%s: %.1f ms -> %.1f records/s; %d solves (%.1f per record, %d warm), %d launches; outcomes: %s'
          % (T, k, l, N, cov, (t1 - t0) * 1e3, T / (t1 - t0), eng.stats['solves'], eng.stats['solves'] / T,
             eng.stats.get('warm_solves', 0), eng.stats['launches'], {o: oc.count(o) for o in set(oc)}))


if __name__ == '__main__':
    main()
