#!/usr/bin/env python3
"""Robustness sweep (GPU): random small orders, point counts and batch sizes through Interpolate.fit_records and
Estimate.evaluate_coeffs.  Checks self-consistency only (no crash, chi^2 recomputed from the returned coefficients, NaN
rows exactly where alpha is NaN, evaluation = basis @ C) - parity is the business of tests/."""
import io
import os
import sys
import tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth
from volumetricinterp_amd.interpolate import Interpolate
from volumetricinterp_amd.estimate import Estimate

CFG = """[DEFAULT]
PARAM = dens
FILENAME = none.h5
OUTPUTFILENAME = none_out.h5
REGULARIZATION_LIST = %s
REGULARIZATION_METHOD = chi2
ERRLIM = 1e9,1e13
GOODFITCODE = 1,2,3,4
CHI2LIM = 0.1,10

[MODEL]
NAME = sphharmlag
MAXK = %d
MAXL = %d
CAP_LIM = %g
MAX_Z_INT = INF
LATCP = 78
LONCP = 262
"""


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    d = tempfile.mkdtemp()
    for case in range(ncase):
        maxk, maxl = int(rng.integers(1, 6)), int(rng.integers(1, 5))
        reg = ['curvature', '0thorder'][int(rng.integers(0, 2))]
        cap = [10., 15., 12.7][int(rng.integers(0, 3))]
        nb, nr = int(rng.integers(2, 12)), int(rng.integers(5, 60))
        T = int(rng.integers(1, 6)) if case % 3 else int(rng.integers(8, 48))          # batches: shared walk, refine
        p = os.path.join(d, 'c%d.ini' % case)
        open(p, 'w').write(CFG % (reg, maxk, maxl, cap))
        it = Interpolate(p)
        N = it.model.nbasis
        lat, lon, alt = synth.beams(nb, nr, seed=case)
        A = it.model.basis(lat, lon, alt)
        if not np.all(np.isfinite(A)):
            print('case %d: order (%d,%d) cap %g: basis has non-finite columns, skipped' % (case, maxk, maxl, cap))
            continue
        value, error = synth.synth_records(A, T, seed0=100 * case)
        if T > 1:
            value[T - 1, :3] = np.nan
        R = it.model.eval_reg_matricies[reg]()
        res = it.fit_records(lat, lon, alt, value, error, {reg: R})
        bad = 0
        for t in range(T):
            a = res['reg_params'][t][reg]
            C = res['Coeffs'][t]
            if np.isnan(a):
                assert np.all(np.isnan(C)), (case, t)
                bad += 1
                continue
            assert np.all(np.isfinite(C)), (case, t)
            fin = np.isfinite(value[t])
            chi = np.sum((A[fin] @ C - value[t][fin])**2 * error[t][fin]**-2)
            assert abs(chi - res['chi_sq'][t]) <= 1e-6 * max(chi, 1e-300), (case, t, chi, res['chi_sq'][t])
        Cg = np.nan_to_num(res['Coeffs'])
        es = Estimate.from_arrays(Cg, None, synth.unix_times(T), np.zeros((4, 3)), CFG % (reg, maxk, maxl, cap))
        q = int(rng.integers(1, 700))
        out = es.evaluate_coeffs(Cg, lat[:q], lon[:q], alt[:q], check_hull=False)
        ref = Cg @ A[:q].T
        err = np.max(np.abs(out - ref)) / max(np.max(np.abs(ref)), 1e-300)
        assert err <= 1e-9, (case, err)
        print('case %2d: order (%d,%d) N=%3d cap %4.1f %-9s P=%4d T=%d: ok (%d NaN rows, eval err %.1e)' %
              (case, maxk, maxl, N, cap, reg, lat.size, T, bad, err), flush=True)
    print('fuzz ok')


if __name__ == '__main__':
    main()
