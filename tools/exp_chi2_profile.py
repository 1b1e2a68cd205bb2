#!/usr/bin/env python3
"""Diagnostic: chi^2(alpha) - nu sampled densely over the bracket the search of the bench record walks into."""
import io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth
from volumetricinterp_amd.fitengine import FitEngine
from volumetricinterp_amd.models.sphharmlag import Model
from tools.perf_fit import CFG


def main():
    lo, hi, K = float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
    m = Model(io.StringIO(CFG % (4, 6)))
    ctx = m.ctx
    lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
    P, N = lat.size, m.nbasis
    d = [ctx.to_device(a) for a in (lat, lon, alt)]
    At = m.basis_device(d[0], d[1], d[2], P, transposed=True)
    A = At.download().T
    R = m.eval_reg_matricies['curvature']()
    value, error = synth.synth_records(A, 1, seed0=1000)
    eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
    eng.load_records(error**-2., value)
    xs = np.linspace(lo, hi, K)
    al = {'curvature': np.power(10., xs)}
    chi = eng.chi2_batch(np.zeros(K, dtype=np.int32), al)
    f = chi - 0.6 * P
    sg = np.sign(f)
    ch = np.nonzero(sg[1:] != sg[:-1])[0]
    print('sign changes at', [(xs[i], xs[i + 1]) for i in ch])
    for i in range(K):
        print('%.9f %+.6e' % (xs[i], f[i]))


if __name__ == '__main__':
    main()
