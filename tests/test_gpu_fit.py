"""GPU parity of the fit path (normal equations, truncated solves, alpha search, covariance) through the
C-ABI, stage by stage (SURVEY 8c ladder L3..L7) against the reference's golden vectors and the oracle."""
import ctypes as C
import io
import math
import os
import warnings

import numpy as np
import pytest
import scipy.linalg

from conftest import load_golden, rel

pytestmark = pytest.mark.gpu


def make_interp(tmp_path, cfg_text):
    from volumetricinterp_amd.interpolate import Interpolate
    p = os.path.join(str(tmp_path), 'config.ini')
    with open(p, 'w') as f:
        f.write(cfg_text)
    return Interpolate(p)


def reg_of(f):
    reg = str(f['reg']) if 'reg' in f.files else None
    return ({reg: f['R']} if reg else {}), reg


def test_L3_normal_equations_on_reference_A():
    from volumetricinterp_amd import _lib
    from volumetricinterp_amd.fitengine import FitEngine
    f = load_golden('fit_k8l2')
    ctx = _lib.get_context()
    eng = FitEngine.from_host_basis(ctx, f['rec0_A'], {'curvature': f['R']}, ['curvature'])
    W = f['error'][0]**-2
    eng.load_records(np.stack([W, 2 * W]), np.stack([f['value'][0], f['value'][0]]))
    AWA, y = eng.normal_equations()
    assert rel(AWA[0], f['rec0_AWA']) <= 1e-13
    assert rel(y[0], f['rec0_y']) <= 1e-13
    assert rel(AWA[1], 2 * f['rec0_AWA']) <= 1e-13


def solve_direct(X, y, want_H=False):
    from volumetricinterp_amd import _lib, fitengine  # noqa: F401 (registers signatures)
    ctx = _lib.get_context()
    B, N = X.shape[0], X.shape[1]
    dX, dy = ctx.to_device(X), ctx.to_device(y)
    dC, drank = ctx.empty((B, N)), ctx.empty((B,), np.int32)
    dH = ctx.empty((B, N, N)) if want_H else None
    eps = np.finfo(float).eps
    _lib.check(_lib.lib.vi_solve_trunc_f64(ctx.handle, B, N, dX.ptr, dy.ptr, None, eps, dC.ptr, drank.ptr, N * eps,
                                           dH.ptr if want_H else None), 'vi_solve_trunc_f64')
    return dC.download(), drank.download(), (dH.download() if want_H else None)


def test_L4_solve_on_reference_system():
    f = load_golden('fit_k8l2')
    C, rank, H = solve_direct(f['rec0_X'][None], f['rec0_y'][None], want_H=True)
    assert rank[0] == 32
    assert rel(C[0], f['Coeffs'][0]) <= 1e-6
    assert rel(H[0], scipy.linalg.pinv(f['rec0_X'])) <= 1e-6
    dC = H[0] @ f['rec0_AWA'] @ H[0]
    assert rel(dC, f['Covariance'][0]) <= 1e-5


def test_truncated_solve_rank_deficient_indefinite():
    """gelsd / pinv semantics on symmetric indefinite, exactly rank-deficient systems: a full-rank
    indefinite 30 x 30 block and a 10-dimensional exact null space, symmetrically permuted (exact)."""
    rng = np.random.default_rng(1)
    B, N, r = 5, 40, 30
    X, Y = [], []
    for i in range(B):
        Q, _ = np.linalg.qr(rng.standard_normal((r, r)))
        lam = np.concatenate([rng.uniform(0.1, 1, 20) * rng.choice([-1, 1], 20), rng.uniform(1e-9, 1e-6, 10)])
        M = (Q * lam) @ Q.T
        M = 0.5 * (M + M.T) * 1e-19                 # magnitude of A^T W A in this problem
        Z = np.zeros((N, N))
        Z[:r, :r] = M
        perm = rng.permutation(N)
        X.append(Z[np.ix_(perm, perm)])
        Y.append(rng.standard_normal(N) * 1e-8)
    X, Y = np.array(X), np.array(Y)
    C, rank, H = solve_direct(X.copy(), Y, want_H=True)
    for i in range(B):
        # exact answer: spectral pseudo-inverse with the null space removed.  (LAPACK gelsd itself reports a
        # spurious singular value ~3.5e-15 sigma_max for some of these matrices and returns rank 31 - its
        # decisions at rcond = eps are artefact-prone, SURVEY F6 - so SciPy is not the yardstick here.)
        lam, V = np.linalg.eigh(X[i])
        keep = np.abs(lam) > 1e-12 * np.abs(lam).max()
        assert keep.sum() == r
        Hx = (V[:, keep] / lam[keep]) @ V[:, keep].T
        assert rank[i] == r
        assert rel(C[i], Hx @ Y[i]) <= 1e-7
        assert rel(H[i], Hx) <= 1e-7


@pytest.mark.parametrize('root', ['auto', 'brent'])
@pytest.mark.parametrize('name', ['fit_k8l2', 'fit_k8l2_c2', 'fit_k8l2_psi'])
def test_L7_fit_records_screened(tmp_path, monkeypatch, name, root):
    """End to end on screened fixtures (reference self-noise < 1e-8): north-star tolerance 1e-6 on the
    coefficients, same alpha, same chi^2, covariance within 1e-5.  'auto' = guarded multisection for few records
    (falls back to Brent on the multi-root brackets these fixtures contain), 'brent' = the reference's iteration
    only."""
    monkeypatch.setenv('VINTERP_ROOT', root)
    f = load_golden(name)
    # curvature fixtures reproduce themselves to 1e-8 in the reference; the 0thorder one (Psi, positive semidefinite:
    # chi^2 monotone, one root) to 2e-7, still inside the gate
    assert np.all(f['self_noise'] < (1e-6 if name.endswith('psi') else 1e-8))
    regm, reg = reg_of(f)
    it = make_interp(tmp_path, str(f['cfg']))
    res = it.fit_records(f['lat'], f['lon'], f['alt'], f['value'], f['error'], regm)
    for t in range(f['value'].shape[0]):
        assert rel(res['Coeffs'][t], f['Coeffs'][t]) <= 1e-6, t
        assert rel(res['Covariance'][t], f['Covariance'][t]) <= 1e-5, t
        assert abs(res['chi_sq'][t] - f['chi_sq'][t]) <= 1e-6 * f['chi_sq'][t]
        assert abs(math.log10(res['reg_params'][t][reg]) - math.log10(f['alpha'][t])) <= 1e-7
    if root == 'brent':
        # work accounting: one solve per distinct alpha (memoised, plus walk prefetch) - far fewer than the reference
        assert it.fit_stats['solves'] < int(f['evalC_calls'])


SCREENED_R4 = ['k4l2', 'k6l2', 'k12l2', 'k6l2_c2', 'k12l2_c2', 'k8l2_c5']


@pytest.mark.parametrize('tag', SCREENED_R4)
def test_L7_screened_orders_and_geometries(tmp_path, tag):
    """VERDICT round 3 item 6 - wider ground for the 1e-6 gate.  tools/gen_golden.py (gen_screened) ran the screening rule of
    SURVEY 8c over thirteen more orders / geometries with the reference itself (three runs each: as is, and twice with 1e-14
    relative noise on its basis; tests/golden/screening.npz keeps the outcome of all thirteen).  The six that reproduce
    themselves to better than 1e-7 - MAXL 2 with MAXK 4, 6, 12 at 11 x 50, MAXK 6 and 12 at 26 x 100, MAXK 8 at 64 x 200
    (12 800 points per record), curvature regularisation by the reference's own Omega of that order - are gated here at the
    north-star tolerance: coefficients and densities (8^3 grid) 1e-6, covariance 1e-5, chi^2 1e-6, log10 alpha 1e-6, and the
    records the reference ends without a root (NaN rows, interpolate.py:558-563) are NaN rows here.  What failed the screen
    and is therefore NOT gated at 1e-6: MAXK 16 x MAXL 2 (7e-7), every 0thorder case at MAXL 3 (1e-3 .. 1e-2),
    MAXK 12 x MAXL 2 0thorder (5e-7), the 64 x 200 0thorder case (1.2e-6), RBF with 27 centres at 26 x 100 (2e-6)."""
    from volumetricinterp_amd import synth
    from volumetricinterp_amd.estimate import Estimate
    f = load_golden('fit_scr_' + tag)
    regm, reg = reg_of(f)
    has = np.all(np.isfinite(f['Coeffs']), axis=1)
    assert has.any() and np.nanmax(f['self_noise']) < 1e-7
    it = make_interp(tmp_path, str(f['cfg']))
    res = it.fit_records(f['lat'], f['lon'], f['alt'], f['value'], f['error'], regm)
    g = synth.query_grid(8)
    es = Estimate.from_arrays(f['Coeffs'], f['Covariance'], f['utime'], f['hull_vert'], str(f['cfg']))
    worst = 0.
    for t in range(f['value'].shape[0]):
        if not has[t]:
            assert np.all(np.isnan(res['Coeffs'][t])) and np.isnan(res['chi_sq'][t]), t
            continue
        worst = max(worst, rel(res['Coeffs'][t], f['Coeffs'][t]))
        assert rel(res['Coeffs'][t], f['Coeffs'][t]) <= 1e-6, t
        assert rel(res['Covariance'][t], f['Covariance'][t]) <= 1e-5, t
        assert abs(res['chi_sq'][t] - f['chi_sq'][t]) <= 1e-6 * f['chi_sq'][t]
        assert abs(math.log10(res['reg_params'][t][reg]) - math.log10(f['alpha'][t])) <= 1e-6
        d_ref = es.evaluate_coeffs(f['Coeffs'][t:t + 1], *g, check_hull=False)[0]
        d_gpu = es.evaluate_coeffs(res['Coeffs'][t:t + 1], *g, check_hull=False)[0]
        assert rel(d_gpu, d_ref) <= 1e-6, t
    print('fit_scr_%s: worst rel(C) vs the reference %.2e (reference self-noise %.1e)' % (tag, worst, np.nanmax(f['self_noise'])))


@pytest.mark.parametrize('regmat_mode', ['default', 'gauss'])
def test_L7_fit_with_the_build_s_own_psi(tmp_path, monkeypatch, regmat_mode):
    """VERDICT round 3, missing #5: every 0thorder parity test fed the fixture's R.  Here the regularisation matrix is the one
    the PRODUCT computes (Model.eval_reg_matricies['0thorder'](), as Interpolate.calc_coeffs does, interpolate.py:496-498) and
    the result is held to the reference's coefficients of tests/golden/fit_k8l2_psi.npz at the north-star 1e-6.
    'default' = the QUADPACK restatement (bit-identical to the reference's Psi - asserted), 'gauss' = the opt-in exact
    quadrature, 5e-10 of max|Psi| away: its effect on alpha, chi^2 and the coefficients is measured here and stays inside the
    same gates (so the opt-in is harmless on this fixture; the default is the reference's numbers regardless)."""
    if regmat_mode == 'gauss':
        monkeypatch.setenv('VINTERP_REGMAT', 'gauss')
    else:
        monkeypatch.delenv('VINTERP_REGMAT', raising=False)
    f = load_golden('fit_k8l2_psi')
    it = make_interp(tmp_path, str(f['cfg']))
    R = it.model.eval_reg_matricies['0thorder']()
    if regmat_mode == 'default':
        assert np.array_equal(R, f['R'])
    else:
        d = np.max(np.abs(R - f['R'])) / np.max(np.abs(f['R']))
        assert 0 < d <= 5e-10
    res = it.fit_records(f['lat'], f['lon'], f['alt'], f['value'], f['error'], {'0thorder': R})
    worst = 0.
    for t in range(f['value'].shape[0]):
        worst = max(worst, rel(res['Coeffs'][t], f['Coeffs'][t]))
        assert rel(res['Coeffs'][t], f['Coeffs'][t]) <= 1e-6, t
        assert rel(res['Covariance'][t], f['Covariance'][t]) <= 1e-5, t
        assert abs(res['chi_sq'][t] - f['chi_sq'][t]) <= 1e-6 * f['chi_sq'][t]
        assert abs(math.log10(res['reg_params'][t]['0thorder']) - math.log10(f['alpha'][t])) <= 1e-7
    print('own Psi (%s): worst rel(C) vs the reference %.2e' % (regmat_mode, worst))


def test_small_order_root_does_not_depend_on_the_batch(tmp_path):
    """Below N = 100 up to four records are searched by the guarded multisection (63 samples per round whatever the number
    of records), larger batches by Brent alone.  Records fitted alone, in pairs and in fours get the same answer bit for
    bit; in a batch of eight (Brent) the same root to 1e-7 decades - both finders stop at that precision on a function
    whose own reproducibility in the reference is 1e-8 here - and coefficients within the north-star tolerance."""
    from volumetricinterp_amd import synth
    f = load_golden('fit_k8l2')
    regm, reg = reg_of(f)
    it = make_interp(tmp_path, str(f['cfg']))
    A = it.model.basis(f['lat'], f['lon'], f['alt'])
    value, error = synth.synth_records(A, 8, seed0=7000)
    fits = {n: it.fit_records(f['lat'], f['lon'], f['alt'], value[:n], error[:n], regm) for n in (1, 2, 4, 8)}
    for n in (2, 4):
        a1, a2 = fits[1]['reg_params'][0][reg], fits[n]['reg_params'][0][reg]
        assert a1 == a2 or (np.isnan(a1) and np.isnan(a2)), (n, a1, a2)
        assert np.array_equal(fits[1]['Coeffs'][0], fits[n]['Coeffs'][0], equal_nan=True), n
    for t in range(2):
        a2, a4 = fits[2]['reg_params'][t][reg], fits[4]['reg_params'][t][reg]
        assert a2 == a4 or (np.isnan(a2) and np.isnan(a4))
    nroot = 0
    for t in range(4):
        a4, a8 = fits[4]['reg_params'][t][reg], fits[8]['reg_params'][t][reg]
        if np.isnan(a4) or np.isnan(a8) or a4 == 0 or a8 == 0:
            assert (np.isnan(a4) and np.isnan(a8)) or a4 == a8
            continue
        nroot += 1
        assert abs(math.log10(a4) - math.log10(a8)) <= 1e-7, (t, a4, a8)
        assert rel(fits[4]['Coeffs'][t], fits[8]['Coeffs'][t]) <= 1e-6, t
    assert nroot >= 2


class _PerturbedBasis(object):
    """Oracle model whose basis carries 1e-14 relative noise: the screen tools/gen_golden.py applies to the reference
    itself (a record whose fit moves under it is decided by rounding - truncation rank flips at the rcond threshold,
    where LAPACK's tiny singular values are noise - and no implementation can be expected to reproduce it)."""

    def __init__(self, model, seed):
        self._m, self._rng = model, np.random.default_rng(seed)

    def __getattr__(self, name):
        return getattr(self._m, name)

    def basis(self, lat, lon, alt):
        A = self._m.basis(lat, lon, alt)
        return A * (1 + 1e-14 * self._rng.standard_normal(A.shape))


@pytest.mark.parametrize('maxk,maxl,reg,seed', [(8, 2, 'curvature', 80), (8, 2, '0thorder', 81), (8, 2, 'curvature', 82)])
def test_fresh_records_vs_oracle(tmp_path, maxk, maxl, reg, seed):
    """Parity beyond the committed fixtures: fresh geometry and records, GPU fit against the CPU oracle on identical
    inputs.  Every record is first screened the way the fixtures were: the oracle is rerun with 1e-14 relative noise on
    its basis and the record only counts if its coefficients move by s < 1e-6; the GPU must then agree within
    max(1e-6, 10 s) <= 1e-5.  (Measured: at MAXK 8, MAXL 2 s is 1e-9..1e-6; at (2,3) 1e-8..1e-4, one draw of s not bounding the next, so that
    order is not used; at (4,3) and (3,4) 1e-4..1e-2: X(alpha) there has eigenvalues a few eps * lambda_max - 4.96e-16
    kept, 1.95e-16 dropped at the root of record 0 - which LAPACK resolves with absolute error eps * sigma_max, i.e. ~50 %
    relative, and whose 1/lambda dominates |C|; the Jacobi solver resolves them to full relative accuracy, so GPU and
    oracle differ by O(1..10) in |C| while chi^2 agrees.  Such orders are not run here - no record of them passes the
    screen; the default order has its own test, test_gpu_default_order.py.)"""
    import re
    import warnings
    import oracle
    from volumetricinterp_amd import synth
    cfg = str(load_golden('fit_k8l2_psi')['cfg'])
    cfg = re.sub(r'MAXK = \d+', 'MAXK = %d' % maxk, cfg)
    cfg = re.sub(r'MAXL = \d+', 'MAXL = %d' % maxl, cfg)
    cfg = cfg.replace('REGULARIZATION_LIST = 0thorder', 'REGULARIZATION_LIST = ' + reg)
    it = make_interp(tmp_path, cfg)
    assert it.regularization_list == [reg] and it.model.nbasis == maxk * maxl**2
    lat, lon, alt = synth.beams(*synth.GEOM_C1, seed=seed)
    o = oracle.SphHarmLagOracle(maxk=maxk, maxl=maxl)
    A = o.basis(lat, lon, alt)
    T = 8
    value, error = synth.synth_records(A, T, seed0=seed * 10)
    value[2, 5:9] = np.nan                                       # dropped points (interpolate.py:516-520)
    R = it.model.eval_reg_matricies[reg]()
    res = it.fit_records(lat, lon, alt, value, error, {reg: R})
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        C, dC, c2, params = oracle.fit_records(o, lat, lon, alt, value, error, {reg: R}, [reg])
        Cp, _, _, _ = oracle.fit_records(_PerturbedBasis(o, seed), lat, lon, alt, value, error, {reg: R}, [reg])
    checked = 0
    for t in range(T):
        a_ref, a = params[t][reg], res['reg_params'][t][reg]
        if np.isnan(a_ref):
            assert np.isnan(a) and np.all(np.isnan(res['Coeffs'][t])), t
            continue
        sn = rel(Cp[t], C[t]) if np.all(np.isfinite(Cp[t])) else float('inf')
        if not sn < 1e-6:
            print('[order (%d,%d) %s record %d] not screened: oracle self-noise %.1e; GPU vs oracle rel(C) %.1e' %
                  (maxk, maxl, reg, t, sn, rel(res['Coeffs'][t], C[t])))
            continue
        checked += 1
        tol = max(1e-6, 10 * sn)
        if a_ref == 0:
            assert a == 0
        else:
            assert abs(math.log10(a) - math.log10(a_ref)) <= max(1e-7, tol), t
        assert rel(res['Coeffs'][t], C[t]) <= tol, (t, sn)
        assert abs(res['chi_sq'][t] - c2[t]) <= max(1e-6, tol) * c2[t]
    assert checked >= 2, 'fewer than 2 of %d records passed the screen' % T


def test_fit_edge_outcomes(tmp_path):
    f = load_golden('fit_edge')
    regm, reg = reg_of(f)
    it = make_interp(tmp_path, str(f['cfg']))
    res = it.fit_records(f['lat'], f['lon'], f['alt'], f['value'], f['error'], regm)
    assert res['reg_params'][0][reg] == 0 and f['alpha'][0] == 0             # 'too smooth' -> alpha = 0
    assert np.isnan(res['reg_params'][1][reg])                               # no root -> NaN row
    assert np.all(np.isnan(res['Coeffs'][1])) and np.all(np.isnan(res['Covariance'][1])) and np.isnan(res['chi_sq'][1])
    assert np.array_equal(np.isnan(res['Coeffs']), np.isnan(f['Coeffs']))
    assert rel(res['Coeffs'][2], f['Coeffs'][2]) <= 1e-6
    assert abs(res['chi_sq'][0] - f['chi_sq'][0]) <= 1e-6 * f['chi_sq'][0]
    assert abs(res['chi_sq'][2] - f['chi_sq'][2]) <= 1e-6 * f['chi_sq'][2]


def test_nonfinite_weights_give_nan_row(tmp_path):
    f = load_golden('fit_k8l2')
    regm, reg = reg_of(f)
    it = make_interp(tmp_path, str(f['cfg']))
    value, error = f['value'][:2].copy(), f['error'][:2].copy()
    error[1, 5] = 0.0                        # W = inf -> lstsq ValueError -> caught -> NaN row (interpolate.py:142-145)
    res = it.fit_records(f['lat'], f['lon'], f['alt'], value, error, regm)
    assert np.all(np.isnan(res['Coeffs'][1])) and np.isnan(res['chi_sq'][1])
    assert rel(res['Coeffs'][0], f['Coeffs'][0]) <= 1e-6


def test_public_eval_C_and_find_reg_param(tmp_path):
    import oracle
    f = load_golden('fit_k8l2')
    regm, reg = reg_of(f)
    it = make_interp(tmp_path, str(f['cfg']))
    A = f['rec0_A']
    b, W = f['value'][0], f['error'][0]**-2
    rp = it.find_reg_param(A, b, W, regm)
    assert abs(math.log10(rp[reg]) - math.log10(f['alpha'][0])) <= 1e-7
    C, dC = it.eval_C(A, b, W, regm, {reg: f['alpha'][0]}, calccov=True)
    Cref, dCref = oracle.eval_C(A, b, W, regm, {reg: f['alpha'][0]}, [reg], calccov=True)
    assert rel(C, Cref) <= 1e-6 and rel(dC, dCref) <= 1e-5
    assert rel(it.eval_C(A, b, W, regm, {reg: f['alpha'][0]}), Cref) <= 1e-6
    nu = 0.6 * len(b)
    v = it.chi2objfunct(-20., A, b, W, regm, nu, reg)
    vref = oracle.chi2objfunct(-20., A, b, W, regm, nu, reg, [reg])
    assert abs(v - vref) <= 1e-6 * abs(vref + nu)
    with pytest.raises(ValueError):
        it.eval_C(A, b * np.nan, W, regm, {reg: 1e-20})
    assert np.isnan(it.find_reg_param(A, b, W * np.inf, regm)[reg])
    with pytest.raises(NotImplementedError):
        it.find_reg_param(A, b, W, regm, method='manual')


def test_rbf_fit_no_regularisation(tmp_path):
    f = load_golden('fit_rbf')
    it = make_interp(tmp_path, str(f['cfg']))
    assert it.regularization_list == []
    res = it.fit_records(f['lat'], f['lon'], f['alt'], f['value'], f['error'], {})
    # A^T W A has condition number ~1e57 here: the coefficients are decided by where the truncation falls
    # (the reference moves by 5e-6 .. 7e-5 under a 1e-14 perturbation of A, and an SVD and an
    # eigen-decomposition truncate differently) - report-only; chi^2 is well defined and is gated.
    for t in range(2):
        assert abs(res['chi_sq'][t] - f['chi_sq'][t]) <= 1e-6 * f['chi_sq'][t]
        assert rel(res['Coeffs'][t], f['Coeffs'][t]) <= 5e-2


def test_solver_fallback_outside_the_in_lds_range():
    """N = 200 does not fit the in-LDS Jacobi kernel (161 KB > 160 KB LDS): vi_solve_trunc_f64 falls back to
    rocSOLVER syevd on the rescaled system - same truncation semantics, with and without pinv."""
    rng = np.random.default_rng(3)
    B, N = 3, 200
    X, Y = [], []
    for i in range(B):
        A = rng.standard_normal((3 * N, N))
        M = (A.T @ A) * 1e-19 + 10.0**(-21 - i) * np.diag(rng.standard_normal(N))
        X.append(0.5 * (M + M.T))
        Y.append(rng.standard_normal(N) * 1e-8)
    X, Y = np.array(X), np.array(Y)
    C, rank, H = solve_direct(X.copy(), Y, want_H=True)
    C2, rank2, _ = solve_direct(X.copy(), Y, want_H=False)
    for i in range(B):
        ref = scipy.linalg.lstsq(X[i], Y[i])[0]
        assert rank[i] == rank2[i] == N
        assert rel(C[i], ref) <= 1e-9 and rel(C2[i], ref) <= 1e-9
        assert rel(H[i], scipy.linalg.pinv(X[i])) <= 1e-9


def test_gcv_on_gpu_matches_reference(tmp_path):
    """REGULARIZATION_METHOD = gcv: leave-one-out objective as batched rank-one down-dated solves on the device,
    Nelder-Mead on the host (the reference's own scipy call)."""
    f = load_golden('fit_gcv')
    regm, reg = reg_of(f)
    it = make_interp(tmp_path, str(f['cfg']))
    assert it.reg_method == 'gcv'
    # stage: objective values at the alphas the reference evaluated (record 0)
    import oracle
    o = oracle.SphHarmLagOracle.from_config(io.StringIO(str(f['cfg'])))
    A = o.basis(f['lat'], f['lon'], f['alt'])
    W = f['error'][0]**-2
    for a, v in f['gcv_calls'][:4]:
        got = it.gcvobjfunct(a, A, f['value'][0], W, regm, reg)
        assert abs(got - v) <= 1e-6 * abs(v), (a, got, v)
    res = it.fit_records(f['lat'], f['lon'], f['alt'], f['value'], f['error'], regm)
    for t in range(2):
        assert abs(math.log10(res['reg_params'][t][reg]) - math.log10(f['alpha'][t])) <= 1e-5
        assert rel(res['Coeffs'][t], f['Coeffs'][t]) <= 1e-5
        assert abs(res['chi_sq'][t] - f['chi_sq'][t]) <= 1e-5 * f['chi_sq'][t]
    assert abs(math.log10(it.gcv(A, f['value'][0], W, regm, reg)) - math.log10(f['alpha'][0])) <= 1e-5
    assert abs(math.log10(it.find_reg_param(A, f['value'][0], W, regm, method='gcv')[reg])
               - math.log10(f['alpha'][0])) <= 1e-5


@pytest.mark.parametrize('shared', ['1', '0'])
def test_full_batch_walk_paths_keep_parity(tmp_path, monkeypatch, shared):
    """Batches solve the bracket walk in the shared bases of the batch (one reference decomposition per decade, bracket
    ends again from cold solves); with that switched off every walk system is solved cold.  40 records = 10 copies of the
    screened golden records (record 1 brackets in [-31, -30]): same parity gates as L7 on both paths, and the same
    numbers bit for bit (test_gpu_configs.py checks that at the default order)."""
    monkeypatch.setenv('VINTERP_SHAREDWALK', shared)
    f = load_golden('fit_k8l2')
    regm, reg = reg_of(f)
    it = make_interp(tmp_path, str(f['cfg']))
    reps = 10
    value = np.tile(f['value'], (reps, 1))
    error = np.tile(f['error'], (reps, 1))
    res = it.fit_records(f['lat'], f['lon'], f['alt'], value, error, regm)
    if shared == '1':
        assert it.fit_stats.get('shared_solves', 0) > 40 * 30        # the walk really went through the shared bases ...
        assert 0 < it.fit_stats.get('reference_solves', 0) <= 102    # ... of one reference system per decade
    else:
        assert it.fit_stats.get('shared_solves', 0) == 0
    assert it.fit_stats.get('walk_same_system', 0) > 40 * 30         # the far decades are one system, solved once
    T0 = f['value'].shape[0]
    for i in range(reps * T0):
        t = i % T0
        assert rel(res['Coeffs'][i], f['Coeffs'][t]) <= 1e-6, i
        assert rel(res['Covariance'][i], f['Covariance'][t]) <= 1e-5, i
        assert abs(math.log10(res['reg_params'][i][reg]) - math.log10(f['alpha'][t])) <= 1e-7
    # identical inputs -> identical outputs within the batch
    assert rel(res['Coeffs'][4], res['Coeffs'][0]) <= 1e-12
