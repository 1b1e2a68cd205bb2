"""Parity at the default, benchmarked order (MAXK 4, MAXL 6 -> N = 144), with the reference's own behaviour as yardstick.

At this order the reference does not reproduce itself: the curvature matrix is indefinite, chi^2(alpha)
has poles and several roots inside one unit bracket (DESIGN.md section 2 shows a scan), X(alpha) has eigenvalues at the
rcond = eps truncation threshold of scipy.linalg.lstsq, and LAPACK decides which of them survive by rounding noise: one
ulp on alpha moves the reference's own chi^2 by 1e-3 and its coefficients by O(1).  tests/golden/fit_default16.npz
(11 x 50) and fit_default_c2.npz (26 x 100, BASELINE configs[1]) hold, for every record, THREE runs of the reference
(tools/gen_golden.py, gen_default_many): as is, and with 1e-14 relative noise on its basis matrix (two seeds).

Gated (each gate is one a wrong implementation fails and a correct one can meet):
  G1  outcome class: root / alpha = 0 / NaN row, and the chi^2 target nu = scale factor x points, equal to the
      reference's whenever its three runs agree on it;
  G2  self-consistency: chi^2 of the returned coefficients within 1e-6 of nu, or the record is flagged by FitEngine's
      guard (a jump of chi^2(alpha)), never a silent mismatch between search and final solve;
  G3  the returned alpha is a root of the REFERENCE's objective: the oracle's chi^2 (LAPACK gelsd on the host, the
      reference's own arithmetic) at the GPU's alpha meets nu within the oracle's own evaluation noise band;
  G4  same alpha, same answer: the GPU densities in the hull against the oracle's densities from eval_C at the GPU's
      alpha, relative to the spread of the oracle itself under a one-ulp change of that alpha.
Reported, not gated: |dlog10 alpha| and density deviation from the unperturbed reference run relative to the spread of
its three runs.  That ratio is large whenever the bracket holds several roots and the three reference runs happen to
pick the same one (which root Brent lands on is decided by 1e-3 differences in chi^2 between LAPACK and an accurate
eigen-solver); the GPU path then returns ANOTHER root of the same objective - which G3 verifies.
"""
import math
import os

import numpy as np
import pytest

from conftest import load_golden, rel

pytestmark = pytest.mark.gpu

LOG_FLOOR = 1e-6          # floors of the self-noise yardstick (Brent's xtol in log10 alpha is 2e-12, the chi^2 target
DENS_FLOOR = 1e-6         # is met to ~1e-7; the north-star tolerance is 1e-6)
# gates G3 / G4 per fixture: (G3 median, G3 max, G4 median, G4 max); None = reported only.  Measured values (DESIGN.md
# section 2): 26 x 100: G3 1e-4 .. 4e-2 (the 4e-2 next to a pole of chi^2, where LAPACK and an accurate solver differ by
# tens of per cent), G4 3e-4 .. 9e-3 against a one-ulp spread of the oracle itself of 6e-5 .. 2e-3.  11 x 50 (550 points for
# 144 functions, numerical rank 77): G3 2e-3 .. 3e-1; the densities of LAPACK and of an accurate solver at the SAME alpha
# differ by 0.1 .. 1.8 there - the coefficients are dominated by directions whose singular values LAPACK only knows to
# 10-100 % - so G4 is not a meaningful gate on that geometry.
GATES = {'fit_default16': (5e-2, 5e-1, None, None), 'fit_default_c2': (5e-2, 1e-1, 2e-2, 5e-2)}


def _classes(f, sfx):
    a, nu = f['alpha' + sfx], f['nu' + sfx]
    out = []
    for t in range(len(a)):
        if np.isnan(a[t]):
            out.append(('nan', None))
        elif a[t] == 0:
            out.append(('zero', None))
        else:
            out.append(('root', round(float(nu[t]), 6)))
    return out


def _fit(tmp_path, f):
    from test_gpu_fit import make_interp
    it = make_interp(tmp_path, str(f['cfg']))
    res = it.fit_records(f['lat'], f['lon'], f['alt'], f['value'], f['error'], {'curvature': f['R']})
    return it, res


def _densities(f, Coeffs):
    from volumetricinterp_amd.estimate import Estimate
    from volumetricinterp_amd import synth
    es = Estimate.from_arrays(Coeffs, None, f['utime'], f['hull_vert'], str(f['cfg']))
    g = synth.query_grid(8)
    return [es.evaluate_coeffs(Coeffs[t:t + 1], *g, check_hull=True)[0] for t in range(Coeffs.shape[0])]


@pytest.mark.parametrize('name', ['fit_default16', 'fit_default_c2'])
def test_default_order_against_reference(tmp_path, capsys, name):
    import warnings
    import oracle
    from volumetricinterp_amd import synth
    f = load_golden(name)
    T = f['value'].shape[0]
    it, res = _fit(tmp_path, f)
    info = res['search']['curvature']
    dens = _densities(f, np.nan_to_num(res['Coeffs']))
    ref_cls = [_classes(f, s) for s in ('', '_p1', '_p2')]
    npts = np.isfinite(f['value']).sum(axis=1)
    o = oracle.SphHarmLagOracle()
    A = o.basis(f['lat'], f['lon'], f['alt'])
    Aq = o.basis(*[x.ravel() for x in synth.query_grid(8)])
    regm = {'curvature': f['R']}
    ratios_a, ratios_d, g3, g4, lines, flips = [], [], [], [], [], []
    for t in range(T):
        agree = ref_cls[0][t] == ref_cls[1][t] == ref_cls[2][t]
        a = res['reg_params'][t]['curvature']
        if np.isnan(a):
            mine = ('nan', None)
        elif a == 0:
            mine = ('zero', None)
        else:
            mine = ('root', round(float(info['info'][t]['sf'] * npts[t]), 6))
        same = mine == ref_cls[0][t]
        if agree and not same:                                                    # G1
            flips.append((t, mine, ref_cls[0][t]))
        line = '[%s rec %2d] classes ref %s build %s' % (name, t, sorted(set(c[t] for c in ref_cls)), mine)
        if mine[0] == 'root':
            i_t = info['info'][t]
            nu = mine[1]
            ok_c = abs(res['chi_sq'][t] - nu) <= 1e-6 * nu                        # G2
            assert ok_c == i_t.get('consistent'), (t, i_t)
            assert ok_c or abs(i_t['chi2_minus_nu']) > 0, (t, i_t)
            # G3 / G4: the reference's own arithmetic at the GPU's alpha (and one / two ulps beside it)
            fin = np.isfinite(f['value'][t])
            At, bt, Wt = A[fin], f['value'][t][fin], f['error'][t][fin]**-2.
            rd = f['dens'][t].ravel()
            ok = np.isfinite(rd) if np.all(np.isfinite(f['Coeffs'][t])) else np.isfinite(dens[t].ravel())
            chis, dd = [], []
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                for k in range(3):
                    Co = oracle.eval_C(At, bt, Wt, regm, {'curvature': a * (1 + k * 4.5e-16)}, ['curvature'])
                    chis.append(float(sum((At @ Co - bt)**2 * Wt)))
                    dd.append((Aq @ Co)[ok])
            dev3 = min(abs(c - nu) for c in chis) / nu
            band = (max(chis) - min(chis)) / nu
            g3.append(dev3)
            d_same = rel(dens[t].ravel()[ok], dd[0])
            y_same = max(rel(dd[1], dd[0]), rel(dd[2], dd[0]), DENS_FLOOR)
            g4.append((d_same, y_same))
            line += ' | G2 chi2-nu %+.2e%s | G3 oracle chi2(alpha_gpu) off nu by %.1e (its ulp band %.1e) | G4 dens vs oracle at ' \
                    'alpha_gpu %.1e (oracle ulp spread %.1e)' % (res['chi_sq'][t] - nu, '' if ok_c else ' [jump]', dev3, band,
                                                               d_same, y_same)
        if agree and same and mine[0] == 'root':
            rd1, rd2 = f['dens_p1'][t].ravel(), f['dens_p2'][t].ravel()
            okr = np.isfinite(rd)
            la = [math.log10(f['alpha' + s][t]) for s in ('', '_p1', '_p2')]
            s_a = max(abs(la[1] - la[0]), abs(la[2] - la[0]), LOG_FLOOR)
            s_d = max(rel(rd1[okr], rd[okr]), rel(rd2[okr], rd[okr]), DENS_FLOOR)
            d_a = abs(math.log10(a) - la[0])
            d_d = rel(dens[t].ravel()[okr], rd[okr])
            assert np.array_equal(np.isfinite(dens[t].ravel()), okr)             # same hull mask
            ratios_a.append(d_a / s_a)
            ratios_d.append(d_d / s_d)
            line += ' | vs reference run: dlog10a %.1e (3-run spread %.1e)  dens %.1e (spread %.1e)' % (d_a, s_a, d_d, s_d)
        lines.append(line)
    g4d = np.array([x[0] for x in g4])
    g4r = np.array([x[0] / x[1] for x in g4])
    with capsys.disabled():
        print()
        for line in lines:
            print(line)
        print('[%s] G1 class flips %s | G3 oracle-objective miss: median %.1e max %.1e | G4 density vs oracle at the same alpha: '
              'median %.1e max %.1e (x oracle ulp spread: median %.1f max %.1f) | reported: deviation from the reference run '
              'over its 3-run spread, log10 alpha median %.1f max %.1f, density median %.1f max %.1f | redone cold %s'
              % (name, flips, np.median(g3), np.max(g3), np.median(g4d), np.max(g4d), np.median(g4r), np.max(g4r),
                 np.median(ratios_a), np.max(ratios_a), np.median(ratios_d), np.max(ratios_d), info.get('redone_cold')))
    assert not flips, flips                                                        # G1
    assert len(g3) >= T // 2
    g3m, g3x, g4m, g4x = GATES[name]
    assert np.median(g3) <= g3m and np.max(g3) <= g3x                              # G3
    if g4m is not None:
        assert np.median(g4d) <= g4m and np.max(g4d) <= g4x                        # G4


def test_guard_redoes_inconsistent_record_cold(tmp_path, monkeypatch):
    """golden fit_default record 1 (the round-1 finding): the warm search declared chi^2 = nu at an alpha where the cold
    final solve gives 497.42.  With the guard the record is redone cold and equals the VINTERP_WARM=0 result."""
    f = load_golden('fit_default')
    it, res = _fit(tmp_path, f)
    info = res['search']['curvature']
    monkeypatch.setenv('VINTERP_WARM', '0')
    it2, res2 = _fit(tmp_path, f)
    for t in range(2):
        i_t = info['info'][t]
        if info['outcomes'][t] != 'root':
            continue
        nu = i_t['sf'] * 550
        assert i_t['consistent'] == (abs(res['chi_sq'][t] - nu) <= 1e-6 * nu)
        if t in info['redone_cold']:
            assert res['reg_params'][t]['curvature'] == res2['reg_params'][t]['curvature']
            assert np.array_equal(res['Coeffs'][t], res2['Coeffs'][t])


def test_cold_solve_against_exact_arithmetic():
    """The definition the reference implements - minimum-norm solution with the singular values below eps * sigma_max
    dropped (scipy.linalg.lstsq at interpolate.py:462) - evaluated in 50-digit arithmetic (tools/gen_exact.py ->
    tests/golden/exact_default_c2.npz) on the reference's own default-order systems of the 26 x 100 geometry.  The GPU solve
    must give the same rank and chi^2; LAPACK itself misses these values by 1e-3 (its chi^2 is two-valued under one ulp on
    alpha).  The coefficients are compared through what they are used for - A c on the data points is what chi^2 sums -
    because |c| is dominated by the directions next to the cut, where 1 / lambda amplifies any difference."""
    from volumetricinterp_amd import _lib, fitengine  # noqa: F401
    import oracle
    e = load_golden('exact_default_c2')
    f = load_golden('fit_default_c2')
    o = oracle.SphHarmLagOracle()
    A = o.basis(f['lat'], f['lon'], f['alt'])
    ctx = _lib.get_context()
    B, N = e['X'].shape[0], e['X'].shape[1]
    dX, dy = ctx.to_device(e['X'].copy()), ctx.to_device(e['y'])
    dC, drank = ctx.empty((B, N)), ctx.empty((B,), np.int32)
    eps = np.finfo(float).eps
    _lib.check(_lib.lib.vi_solve_trunc_f64(ctx.handle, B, N, dX.ptr, dy.ptr, None, eps, dC.ptr, drank.ptr, N * eps, None),
               'vi_solve_trunc_f64')
    C, rank = dC.download(), drank.download()
    for i in range(B):
        t = int(e['record'][i])
        b, W = f['value'][t], f['error'][t]**-2.
        chi = float(sum((A @ C[i] - b)**2 * W))
        fit_e, fit_g = A @ e['C'][i], A @ C[i]
        print('record %d log10 alpha %.4f: rank %d (exact %d)  chi2 %.6f (exact %.6f, rel %.1e)  rel(A c) %.1e  rel(c) %.1e'
              % (t, e['log10_alpha'][i], rank[i], e['rank'][i], chi, e['chi2'][i], abs(chi - e['chi2'][i]) / e['chi2'][i],
                 rel(fit_g, fit_e), rel(C[i], e['C'][i])))
        edge = np.min(np.abs(e['around_cut'][i] / eps - 1.0))
        if edge < 0.02:
            # an eigenvalue within 2 % of the cut (2.2117e-16 against eps = 2.2204e-16 in one of the systems): which side
            # it falls on is decided by the last digits of max|lambda|; both answers are right
            continue
        assert rank[i] == e['rank'][i]
        assert abs(chi - e['chi2'][i]) <= 1e-4 * e['chi2'][i]          # LAPACK: 1e-3 .. 2e-2 on the same systems
        assert rel(fit_g, fit_e) <= 1e-4
