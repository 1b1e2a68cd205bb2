"""Parity at the default, benchmarked order (MAXK 4, MAXL 6 -> N = 144), with the reference's own behaviour as yardstick.

At this order the reference does not reproduce itself: the curvature matrix is indefinite, chi^2(alpha)
has poles and several roots inside one unit bracket, and X(alpha) has eigenvalues at the rcond = eps truncation threshold
of scipy.linalg.lstsq, which LAPACK resolves to 10-100 %.  What the reference returns therefore depends on things its
source does not choose; tests/golden holds THE REFERENCE ITSELF run 35 times per record (tools/gen_golden.py):
  fit_default16 / fit_default_c2   3 runs: as is, and with 1e-14 relative noise on its basis (two seeds);
  fit_default_roots                16 more runs with 1e-14 noise (seeds 100-115)             - 19 runs with SciPy's
                                   default LAPACK driver of scipy.linalg.lstsq, gelsd (divide and conquer);
  fit_default_drivers              8 runs each (as is + 7 noise seeds) with scipy.linalg.lstsq switched to gelss
                                   (QR-iteration SVD) and to gelsy (complete orthogonal factorisation): the same SciPy
                                   function, the same definition (minimum norm, rcond = eps), another LAPACK routine.
Findings these fixtures encode: gelsd runs cluster within 1e-4 .. 1e-2 decades of log10 alpha and
land on 1-6 different roots per record; the three drivers disagree by 0.01-0.16 decades at 26 x 100 and by up to THREE
decades at 11 x 50 (gelsd -41.8, gelss -38.8, gelsy -39.9 on one record), with densities differing by O(1).

Gated, per record (each gate is one a wrong implementation fails and a correct one can meet):
  G1  outcome class (root / alpha = 0 / NaN row) and chi^2 target nu = scale factor x points equal to the reference's
      whenever its 35 runs agree, and to that of one of its runs otherwise;
  G2  self-consistency: chi^2 of the returned coefficients within 1e-6 of nu, or the record is flagged by FitEngine's
      guard (band / jump / polished), never a silent mismatch between search and final solve; a record flagged 'jump'
      must show that jump in an independent accurate CPU evaluation of the objective (oracle/accurate.py);
  G3  at the GPU's alpha that accurate CPU evaluation gives the GPU's chi^2 (1e-5 nu): the returned alpha is a root (or a
      sign-changing jump) of the reference's DEFINITION evaluated exactly - LAPACK's own value there is 0.1-4 % off;
  R1  the GPU's unit bracket [floor(log10 alpha), +1] is one the reference visited in one of its runs;
  R2  ROOT SET: the GPU's log10 alpha lies within 3 x the spread of a cluster of reference runs (same driver, roots within
      0.02 decades of each other, >= 3 runs) of the nearest member of that cluster;
  R3  the GPU's densities on the 8^3 grid differ from those of that nearest run by at most 4 x the spread of the
      densities among the cluster's runs.
Measured (MI355X, this build): R2 and R3 hold on all 19 root records - for the cluster of the reference's gelss runs (one
record: its gelsd runs at that root), with distances of 2e-6 .. 2e-3 decades against spreads of 7e-6 .. 2e-2 (ratio
<= 0.92) and density ratios <= 2.7.  The
solver of this build is accurate where LAPACK is not (chi^2 within 1e-10 of 50-digit arithmetic,
test_cold_solve_against_exact_arithmetic); of the three LAPACK routines gelss comes closest to that, and the GPU path
is indistinguishable from the reference run on it.  Against the reference's default driver the distances are
1e-5 .. 3 decades - as large as gelsd's own distance from gelss and gelsy - and are reported, not gated.
"""
import math
import os

import numpy as np
import pytest

from conftest import load_golden, rel

pytestmark = pytest.mark.gpu

SPREAD_FLOOR_A = 1e-6        # decades: Brent's xtol is 2e-12, the chi^2 target is met to ~1e-7
SPREAD_FLOOR_D = 1e-6        # relative density: the north-star tolerance
CLUSTER_WIDTH = 0.02         # decades: reference runs whose roots lie closer than this belong to one root
R2_FACTOR, R3_FACTOR = 3.0, 4.0


def _reference_runs(name):
    """All reference runs of a fixture: list of dicts(driver, alpha[T], nu[T], dens[T, ...])."""
    tag = name[len('fit_'):]
    f, r, d = load_golden(name), load_golden('fit_default_roots'), load_golden('fit_default_drivers')
    runs = [dict(driver='gelsd', alpha=f['alpha' + s], nu=f['nu' + s], dens=f['dens' + s]) for s in ('', '_p1', '_p2')]
    runs += [dict(driver='gelsd', alpha=r[tag + '_alpha'][k], nu=r[tag + '_nu'][k], dens=r[tag + '_dens'][k])
             for k in range(r[tag + '_alpha'].shape[0])]
    for drv in ('gelss', 'gelsy'):
        runs += [dict(driver=drv, alpha=d['%s_%s_alpha' % (tag, drv)][k], nu=d['%s_%s_nu' % (tag, drv)][k],
                      dens=d['%s_%s_dens' % (tag, drv)][k]) for k in range(d['%s_%s_alpha' % (tag, drv)].shape[0])]
    return f, runs


def _class(a, nu):
    if np.isnan(a):
        return ('nan', None)
    if a == 0:
        return ('zero', None)
    return ('root', round(float(nu), 6))


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


def _verdict(info, t):
    i_t = info['info'][t]
    if i_t.get('consistent'):
        return 'consistent'
    if i_t.get('jump'):
        return 'jump'
    if i_t.get('redone_cold'):
        return 'redone_cold'
    if t in info.get('polished_cold', []):
        return 'polished'
    return 'band'


@pytest.mark.parametrize('name', ['fit_default16', 'fit_default_c2'])
def test_default_order_against_reference_root_set(tmp_path, capsys, name):
    import oracle
    from oracle import accurate
    f, runs = _reference_runs(name)
    T = f['value'].shape[0]
    it, res = _fit(tmp_path, f)
    info = res['search']['curvature']
    dens = _densities(f, np.nan_to_num(res['Coeffs']))
    npts = np.isfinite(f['value']).sum(axis=1)
    o = oracle.SphHarmLagOracle()
    A = o.basis(f['lat'], f['lon'], f['alt'])
    lines, flips, out_of_bracket, r2, r3, g3, jumps, dist_gelsd, nroot = [], [], [], [], [], [], [], [], 0
    for t in range(T):
        classes = [_class(r_['alpha'][t], r_['nu'][t]) for r_ in runs]
        a = res['reg_params'][t]['curvature']
        mine = _class(a, info['info'][t]['sf'] * npts[t] if (not np.isnan(a) and a != 0) else 0.)
        if mine not in classes:                                                    # G1
            flips.append((t, mine, sorted(set(classes), key=str)))
        if len(set(classes)) == 1 and mine != classes[0]:
            flips.append((t, mine, classes[0]))
        line = '[%s rec %2d] classes ref %s build %s' % (name, t, sorted(set(classes), key=str), mine)
        if mine[0] != 'root':
            lines.append(line)
            continue
        nroot += 1
        nu, la = mine[1], math.log10(a)
        i_t = info['info'][t]
        v = _verdict(info, t)
        ok_c = abs(res['chi_sq'][t] - nu) <= 1e-6 * nu                              # G2
        assert ok_c == bool(i_t.get('consistent')), (t, i_t)
        assert ok_c or abs(i_t['chi2_minus_nu']) > 0, (t, i_t)
        line += ' | G2 %s chi2-nu %+.2e' % (v, res['chi_sq'][t] - nu)
        # G3: an independent, accurate CPU evaluation of the reference's definition (oracle/accurate.py: QR-preconditioned
        # cyclic Jacobi in NumPy, 1e-13 against 50-digit arithmetic) at the GPU's alpha
        fin = np.isfinite(f['value'][t])
        At, bt, Wt = A[fin], f['value'][t][fin], f['error'][t][fin]**-2.
        if v != 'jump':
            c_acc = accurate.chi2_accurate(At, bt, Wt, f['R'], a)[0]
            g3.append(abs(c_acc - res['chi_sq'][t]) / nu)
            line += ' | G3 accurate CPU chi2 at alpha_gpu differs from the GPU\'s by %.1e nu' % g3[-1]
        else:
            # a jump: chi^2(alpha) is discontinuous at the GPU's alpha - an eigenvalue of X(alpha) sits AT the cut
            # eps * max|lambda| there (tools/diag_jump.py: ratio 1.0000, rank 94 | 95, chi^2 548.2 | 556.1 on one record), so
            # chi^2 - nu changes sign without a root.  Where exactly the rank flips is decided in the sixth digit of
            # log10 alpha; 1e-4 decades away on either side the accurate evaluation must sit on the two plateaus: they
            # straddle nu, their gap is at least half the GPU's miss, and the GPU's final chi^2 is one of them
            lo_, hi_ = (accurate.chi2_accurate(At, bt, Wt, f['R'], 10.**(la + dl))[0] for dl in (-1e-4, 1e-4))
            line += ' (accurate CPU chi2 1e-4 decades below | above alpha_gpu: %.3f | %.3f, nu %.0f)' % (lo_, hi_, nu)
            jumps.append((t, lo_, hi_, nu, res['chi_sq'][t]))
            assert (lo_ - nu) * (hi_ - nu) < 0, (t, lo_, hi_, nu)
            assert abs(hi_ - lo_) >= 0.5 * abs(res['chi_sq'][t] - nu), (t, lo_, hi_, res['chi_sq'][t], nu)
            assert min(abs(lo_ - res['chi_sq'][t]), abs(hi_ - res['chi_sq'][t])) <= 1e-3 * nu, (t, lo_, hi_, res['chi_sq'][t])
        # R1: bracket
        roots = [(r_['driver'], math.log10(r_['alpha'][t]), k) for k, r_ in enumerate(runs) if r_['alpha'][t] > 0]
        if math.floor(la) not in set(math.floor(x) for _, x, _k in roots):
            out_of_bracket.append(t)
        # R2 / R3: nearest cluster of >= 3 runs of one driver
        best = None
        for drv in ('gelsd', 'gelss', 'gelsy'):
            rs = [(x, k) for dname, x, k in roots if dname == drv]
            if not rs:
                continue
            x0, k0 = min(rs, key=lambda p_: abs(p_[0] - la))
            members = [(x, k) for x, k in rs if abs(x - x0) <= CLUSTER_WIDTH]
            ok = np.isfinite(runs[k0]['dens'][t].ravel())
            assert np.array_equal(np.isfinite(dens[t].ravel()), ok)                # same hull mask
            sa = max(max(x for x, _ in members) - min(x for x, _ in members), SPREAD_FLOOR_A)
            sd = max([rel(runs[k]['dens'][t].ravel()[ok], runs[k0]['dens'][t].ravel()[ok]) for _, k in members if k != k0]
                     + [SPREAD_FLOOR_D])
            da, dd = abs(x0 - la), rel(dens[t].ravel()[ok], runs[k0]['dens'][t].ravel()[ok])
            line += ' | %s: dlog10a %.1e (n %d, spread %.0e) dens %.1e (spread %.0e)' % (drv, da, len(members), sa, dd, sd)
            if drv == 'gelsd':
                dist_gelsd.append(da)
            # the cluster the GPU's answer belongs to: the one in which both ratios are smallest relative to their gates
            if len(members) >= 3 and (best is None or max(da / sa / R2_FACTOR, dd / sd / R3_FACTOR) <
                                      max(best[0] / R2_FACTOR, best[1] / R3_FACTOR)):
                best = (da / sa, dd / sd, drv)
        assert best is not None, (t, roots)
        line += ' | nearest cluster: %s, R2 ratio %.2f, R3 ratio %.2f' % (best[2], best[0], best[1])
        r2.append(best[0])
        r3.append(best[1])
        lines.append(line)
    with capsys.disabled():
        print()
        for line in lines:
            print(line)
        print('[%s] G1 flips %s | R1 records outside every bracket the reference visited: %s | R2 (log10 alpha over cluster '
              'spread) max %.2f | R3 (density over cluster spread) max %.2f | G3 accurate CPU chi2 vs GPU chi2 at the GPU\'s '
              'alpha: max %.1e nu (%d jumps checked for their gap) | reported: distance to the nearest run of the '
              'reference\'s default driver gelsd: median %.1e max %.1e decades | redone cold %s'
              % (name, flips, out_of_bracket, max(r2), max(r3), max(g3), len(jumps), np.median(dist_gelsd), max(dist_gelsd),
                 info.get('redone_cold')))
    assert not flips, flips                                                         # G1
    assert nroot >= T // 2
    assert len(out_of_bracket) <= 0.25 * nroot, out_of_bracket                      # R1
    assert max(r2) <= R2_FACTOR, r2                                                 # R2
    assert max(r3) <= R3_FACTOR, r3                                                 # R3
    assert max(g3) <= 1e-5, g3                         # G3 (measured: <= 5e-10 at 26 x 100, <= 4.9e-6 at 11 x 50)


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
