"""Stage tests of the entries behind the batched regularisation-parameter search (interpolate.py:152-218), each against
NumPy / LAPACK on the same inputs, at the benchmarked order (N = 144, 26 x 100 points):
  vi_reg_floor_f64     below which alpha the walk systems are one and the same matrix;
  vi_basis_solve_f64   the walk solved in the eigenbasis of a reference system (shared bases);
  vi_decompose_f64 / vi_warm_finish_f64   the two phases of vi_warm_prepare_f64;
  vi_warm_rebase_f64   the rotated system moved next to the root;
  vi_chi2_f64          the fixed-order chi^2 kernel, whatever the batch."""
import os

import numpy as np
import pytest

from conftest import load_golden, rel
from test_gpu_configs import _engine, CFG144, EPS

pytestmark = pytest.mark.gpu
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def setup():
    from volumetricinterp_amd import synth
    m, ctx, eng, A, _ = _engine(CFG144, synth.GEOM_C2)
    P = A.shape[0]
    value, error = synth.synth_records(A, 12, seed0=1000)
    W = error**-2.
    eng.load_records(W, value)
    AWA, y = eng.normal_equations()
    from conftest import load_golden
    R = load_golden('regmat')['default_curvature']
    return dict(m=m, ctx=ctx, eng=eng, A=A, P=P, W=W, b=value, AWA=AWA, y=y, R=R, N=A.shape[1])


def _chi2(s, C, t):
    return float(np.sum((s['A'] @ C - s['b'][t])**2 * s['W'][t]))


def test_reg_floor_marks_bit_identical_systems(setup):
    """Below the floor of a record, fl(AWA + alpha R) is AWA itself in every element; a decade above it, it is not."""
    from volumetricinterp_amd import _lib
    s = setup
    ctx, N = s['ctx'], s['N']
    T = s['AWA'].shape[0]
    dfl = ctx.empty((T,))
    _lib.check(_lib.lib.vi_reg_floor_f64(ctx.handle, T, N, s['eng'].dAWA.ptr, s['eng'].R['curvature'].ptr, dfl.ptr), 'floor')
    fl = dfl.download()
    assert np.all(np.isfinite(fl)) and np.all(fl > 0)
    for t in range(T):
        below = 10.**(np.ceil(np.log10(fl[t])) - 1.)
        assert below < fl[t]
        assert np.array_equal(s['AWA'][t] + below * s['R'], s['AWA'][t])
        assert not np.array_equal(s['AWA'][t] + 1e3 * fl[t] * s['R'], s['AWA'][t])
    # the engine's table: every walk decade under the record's floor is served by the solve at the floor decade
    eng = s['eng']
    eng._find_same_below('curvature')
    k = eng._same_below['curvature']
    assert np.all(k <= np.log10(fl)) and np.all(k > np.log10(fl) - 1.0000001)
    assert np.all(k < -30) and np.all(k > -70)          # measured -47 .. -49 on this geometry: half the walk is one system


def test_basis_solve_equals_cold_solve_in_any_basis(setup):
    """(V^T AWA V + alpha V^T R V) c' = V^T y, C = V c' is the system itself in other coordinates: from the eigenbasis of
    a reference system (mean weights) it must give the chi^2 of the cold solve - to the noise of the problem (the walk
    only uses its sign, with a 1e-3 margin) - and a random orthogonal basis must work just as well, only slower."""
    from volumetricinterp_amd import _lib
    s = setup
    ctx, eng, N = s['ctx'], s['eng'], s['N']
    h = ctx.handle
    ks = np.array([0., -10., -22., -26., -30., -40.])
    # reference = mean of the last 6 records, test records 0..3
    ref = np.mean(s['AWA'][6:], axis=0)
    dAWA = ctx.to_device(np.concatenate([s['AWA'][:4], ref[None]]))
    dy = ctx.to_device(np.concatenate([s['y'][:4], s['y'][:1]]))
    dR = ctx.to_device(s['R'])
    K = len(ks)
    dV, dD1, dD2, dyt = ctx.empty((K, N, N)), ctx.empty((K, N, N)), ctx.empty((K, N, N)), ctx.empty((K, N))
    keep = [ctx.to_device(np.full(K, 4, np.int32)), ctx.to_device(10.**ks), ctx.empty((K, N)), ctx.empty((K,), np.int32)]
    _lib.check(_lib.lib.vi_warm_prepare_f64(h, K, N, dAWA.ptr, keep[0].ptr, keep[1].ptr, dR.ptr, dy.ptr, EPS, keep[2].ptr,
                                            keep[3].ptr, dV.ptr, dD1.ptr, dD2.ptr, dyt.ptr), 'prepare')
    V = dV.download()
    for k in range(K):                                   # the reference bases are orthonormal
        assert np.max(np.abs(V[k] @ V[k].T - np.eye(N))) <= 1e-13
    rec = np.repeat(np.arange(4, dtype=np.int32), K)
    bas = np.tile(np.arange(K, dtype=np.int32), 4)
    al = np.tile(10.**ks, 4)
    B = len(rec)
    dC, drk = ctx.empty((B, N)), ctx.empty((B,), np.int32)
    drec, dbas, dal = ctx.to_device(rec), ctx.to_device(bas), ctx.to_device(al)
    _lib.check(_lib.lib.vi_basis_solve_f64(h, B, N, dAWA.ptr, dy.ptr, drec.ptr, dbas.ptr, dal.ptr, dV.ptr, dD2.ptr, EPS, dC.ptr,
                                           drk.ptr, None), 'basis_solve')
    Cs = dC.download()
    dX = ctx.to_device(np.stack([s['AWA'][r] + a * s['R'] for r, a in zip(rec, al)]))
    dCc, drc = ctx.empty((B, N)), ctx.empty((B,), np.int32)
    _lib.check(_lib.lib.vi_solve_trunc_f64(h, B, N, dX.ptr, dy.ptr, drec.ptr, EPS, dCc.ptr, drc.ptr, N * EPS, None), 'cold')
    Cc = dCc.download()
    worst = 0.
    for i in range(B):
        c_s, c_c = _chi2(s, Cs[i], rec[i]), _chi2(s, Cc[i], rec[i])
        worst = max(worst, abs(c_s - c_c) / c_c)
    assert worst <= 2e-4, worst                           # measured 6e-5 at the worst decade (poles of chi^2 nearby)
    # a random orthogonal basis: same answer (D2 must then be formed for it)
    rng = np.random.default_rng(3)
    Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
    Vr = np.ascontiguousarray(Q.T)                        # library layout: row k = basis vector k
    dVr = ctx.to_device(Vr[None])
    dD2r = ctx.to_device((Q.T @ s['R'] @ Q)[None])
    one = ctx.to_device(np.zeros(1, np.int32))
    dC1 = ctx.empty((1, N))
    da1, drk1 = ctx.to_device(np.array([1e-12])), ctx.empty((1,), np.int32)
    _lib.check(_lib.lib.vi_basis_solve_f64(h, 1, N, dAWA.ptr, dy.ptr, one.ptr, one.ptr, da1.ptr, dVr.ptr, dD2r.ptr, EPS, dC1.ptr,
                                           drk1.ptr, None), 'basis_solve')
    import scipy.linalg
    Cl = scipy.linalg.lstsq(s['AWA'][0] + 1e-12 * s['R'], s['y'][0])[0]
    assert abs(_chi2(s, dC1.download()[0], 0) / _chi2(s, Cl, 0) - 1.) <= 1e-6


def test_two_phase_prepare_equals_one_phase(setup):
    """vi_decompose_f64 + vi_warm_finish_f64 (the single record's speculative bracket bases) leave exactly what
    vi_warm_prepare_f64 leaves: C, V, D1, D2, yt bit for bit; the finish of ONE log out of many picks the right one."""
    from volumetricinterp_amd import _lib
    s = setup
    ctx, eng, N = s['ctx'], s['eng'], s['N']
    h = ctx.handle
    rec = np.array([0, 0, 3, 5], np.int32)
    al = 10.**np.array([-26.5, -30.5, -26.5, -27.5])
    B = len(rec)
    drec, dal = ctx.to_device(rec), ctx.to_device(al)
    out1 = [ctx.empty((B, N)), ctx.empty((B,), np.int32), ctx.empty((B, N, N)), ctx.empty((B, N, N)), ctx.empty((B, N, N)),
            ctx.empty((B, N))]
    _lib.check(_lib.lib.vi_warm_prepare_f64(h, B, N, eng.dAWA.ptr, drec.ptr, dal.ptr, eng.R['curvature'].ptr, eng.dy.ptr, EPS,
                                            *[o.ptr for o in out1]), 'prepare')
    logb = int(_lib.lib.vi_rotation_log_bytes(N))
    assert logb % 8 == 0 and logb > 0
    dlog, dnr = ctx.empty((B * (logb // 8),)), ctx.empty((B,), np.int32)
    dC, drk = ctx.empty((B, N)), ctx.empty((B,), np.int32)
    _lib.check(_lib.lib.vi_decompose_f64(h, B, N, eng.dAWA.ptr, drec.ptr, dal.ptr, eng.R['curvature'].ptr, eng.dy.ptr, EPS,
                                         dC.ptr, drk.ptr, dlog.ptr, dnr.ptr), 'decompose')
    assert np.array_equal(dC.download(), out1[0].download()) and np.array_equal(drk.download(), out1[1].download())
    for j in (2, 0, 3):
        o = [ctx.empty((1, N, N)), ctx.empty((1, N, N)), ctx.empty((1, N, N)), ctx.empty((1, N))]
        drj = ctx.to_device(rec[j:j + 1])
        _lib.check(_lib.lib.vi_warm_finish_f64(h, 1, N, dlog.offset_ptr(j * (logb // 8)), dnr.offset_ptr(j), eng.dAWA.ptr,
                                               drj.ptr, eng.R['curvature'].ptr, eng.dy.ptr, *[x.ptr for x in o]), 'finish')
        for a, b in zip(o, out1[2:]):
            assert np.array_equal(a.download()[0], b.download()[j]), j


def test_rebase_moves_the_rotated_system_and_keeps_the_solution(setup):
    """vi_warm_rebase_f64 returns the warm solution at alpha and leaves a rotated system in which X(alpha) is diagonal:
    V stays orthonormal, D1 + alpha D2 is diagonal to rounding, D1 = V^T AWA V, D2 = V^T R V, yt = V^T y, and a warm solve
    next to alpha from the new basis gives the chi^2 of the cold solve."""
    from volumetricinterp_amd import _lib
    s = setup
    ctx, eng, N = s['ctx'], s['eng'], s['N']
    h = ctx.handle
    rec = np.array([0, 3, 5], np.int32)
    x0 = np.array([-26.5, -26.5, -27.5])
    x1 = np.array([-26.37, -26.61, -27.12])                       # where the iterates cluster
    B = len(rec)
    drec = ctx.to_device(rec)
    dC0, drk = ctx.empty((B, N)), ctx.empty((B,), np.int32)
    dV, dD1, dD2, dyt = ctx.empty((B, N, N)), ctx.empty((B, N, N)), ctx.empty((B, N, N)), ctx.empty((B, N))
    dal0 = ctx.to_device(10.**x0)
    _lib.check(_lib.lib.vi_warm_prepare_f64(h, B, N, eng.dAWA.ptr, drec.ptr, dal0.ptr, eng.R['curvature'].ptr,
                                            eng.dy.ptr, EPS, dC0.ptr, drk.ptr, dV.ptr, dD1.ptr, dD2.ptr, dyt.ptr), 'prepare')
    slot = ctx.to_device(np.arange(B, dtype=np.int32))
    dal = ctx.to_device(10.**x1)
    dCw = ctx.empty((B, N))
    _lib.check(_lib.lib.vi_warm_solve_f64(h, B, N, dD1.ptr, dD2.ptr, dyt.ptr, dV.ptr, slot.ptr, dal.ptr, EPS, dCw.ptr, drk.ptr, None),
               'warm')
    dCr = ctx.empty((B, N))
    # record 0 rides along as a plain warm solve (nplain = 1): its rotated system must stay where it is
    keep0 = [x.download()[0].copy() for x in (dV, dD1, dD2, dyt)]
    _lib.check(_lib.lib.vi_warm_rebase_f64(h, B, 1, N, eng.dAWA.ptr, eng.R['curvature'].ptr, eng.dy.ptr, drec.ptr, slot.ptr, dal.ptr,
                                           EPS, dV.ptr, dD1.ptr, dD2.ptr, dyt.ptr, dCr.ptr, drk.ptr, None), 'rebase')
    assert np.array_equal(dCr.download(), dCw.download())          # the solutions returned are the warm solve's
    for a, b in zip(keep0, (dV, dD1, dD2, dyt)):
        assert np.array_equal(a, b.download()[0])
    # now all three re-base (record 0 too)
    _lib.check(_lib.lib.vi_warm_rebase_f64(h, B, 0, N, eng.dAWA.ptr, eng.R['curvature'].ptr, eng.dy.ptr, drec.ptr, slot.ptr, dal.ptr,
                                           EPS, dV.ptr, dD1.ptr, dD2.ptr, dyt.ptr, dCr.ptr, drk.ptr, None), 'rebase')
    V, D1, D2, yt = dV.download(), dD1.download(), dD2.download(), dyt.download()
    for i, t in enumerate(rec):
        Vm = V[i].T                                               # columns = basis vectors
        assert np.max(np.abs(Vm.T @ Vm - np.eye(N))) <= 1e-12
        assert rel(D1[i], Vm.T @ s['AWA'][t] @ Vm) <= 1e-13
        assert rel(D2[i], Vm.T @ s['R'] @ Vm) <= 1e-13
        assert rel(yt[i], Vm.T @ s['y'][t]) <= 1e-13
        Xr = D1[i] + 10.**x1[i] * D2[i]
        off = Xr - np.diag(np.diag(Xr))
        assert np.max(np.abs(off)) <= 1e-12 * np.max(np.abs(np.diag(Xr)))
    # a warm solve 1e-4 decades away, from the new basis, against the cold solve there
    x2 = x1 + 1e-4
    dal2 = ctx.to_device(10.**x2)
    _lib.check(_lib.lib.vi_warm_solve_f64(h, B, N, dD1.ptr, dD2.ptr, dyt.ptr, dV.ptr, slot.ptr, dal2.ptr, EPS, dCw.ptr, drk.ptr, None),
               'warm')
    Cw = dCw.download()
    dX = ctx.to_device(np.stack([s['AWA'][t] + 10.**x * s['R'] for t, x in zip(rec, x2)]))
    dCc = ctx.empty((B, N))
    _lib.check(_lib.lib.vi_solve_trunc_f64(h, B, N, dX.ptr, eng.dy.ptr, drec.ptr, EPS, dCc.ptr, drk.ptr, N * EPS, None), 'cold')
    Cc = dCc.download()
    for i, t in enumerate(rec):
        # the noise of the problem, not of the distance to the basis: X(alpha) has eigenvalues next to the truncation cut
        # here and two routes to the truncated solution differ by up to 6.4e-5 in chi^2 (record 0; 1e-7 on the others)
        assert abs(_chi2(s, Cw[i], t) / _chi2(s, Cc[i], t) - 1.) <= 2e-4, i


def test_chi2_kernel_has_one_summation_order(setup):
    """chi^2 of a coefficient vector: the same bits alone, among 7 others, among 300 and among 3000 (the three launch shapes
    of vi_chi2_f64), and within 1e-13 of NumPy."""
    from volumetricinterp_amd import _lib
    s = setup
    ctx, eng, N, P = s['ctx'], s['eng'], s['N'], s['P']
    rng = np.random.default_rng(5)
    C0 = np.linalg.lstsq(s['A'], s['b'][2], rcond=None)[0]
    vals = []
    for B in (1, 8, 300, 3000):
        Cb = rng.standard_normal((B, N)) * np.abs(C0)
        pos = B // 3
        Cb[pos] = C0
        rec = rng.integers(0, 12, B).astype(np.int32)
        rec[pos] = 2
        dchi = ctx.empty((B,))
        dCb, drec = ctx.to_device(Cb), ctx.to_device(rec)
        _lib.check(_lib.lib.vi_chi2_f64(ctx.handle, B, P, N, eng.At.ptr, dCb.ptr, drec.ptr, eng.dW.ptr, eng.db.ptr, dchi.ptr),
                   'chi2')
        out = dchi.download()
        vals.append(out[pos])
        j = (pos + 1) % B
        assert abs(out[j] / _chi2(s, Cb[j], rec[j]) - 1.) <= 1e-10       # a random vector: A C cancels heavily, in NumPy too
    assert all(v == vals[0] for v in vals), vals
    # against extended precision (the least-squares coefficients reach 1e20 and A C cancels to 1e11: float64 NumPy is 3e-13
    # away from the kernel, and as far from the truth)
    ld = np.longdouble
    truth = float(np.sum((s['A'].astype(ld) @ C0.astype(ld) - s['b'][2].astype(ld))**2 * s['W'][2].astype(ld)))
    assert abs(vals[0] / truth - 1.) <= 1e-11


def test_qr_similarity_preconditioner():
    """vi_qr_similarity_f64 (csrc/vi_qr.hip): the pre-conditioner of the cold solves.  Q is orthogonal, X1 = Q^T X Q with
    X the symmetric matrix made of the lower triangle (what the Jacobi kernel reads), y1 = Q^T y, to rounding - at the
    benchmarked order on the reference's own systems and at small / odd orders (N = 27: the y column shares an octet with
    the last matrix column; N = 36, 50: padded local rows)."""
    from volumetricinterp_amd import _lib, fitengine  # noqa: F401
    ctx = _lib.get_context()
    e = load_golden('exact_default_c2')
    rng = np.random.default_rng(3)
    cases = [(e['X'], e['y'])]
    for N in (27, 32, 36, 50, 96):
        Xs = []
        for _ in range(3):
            Q, _r = np.linalg.qr(rng.standard_normal((N, N)))
            lam = 10.0**rng.uniform(-30, 0, N) * rng.choice([-1, 1], N)
            M = (Q * lam) @ Q.T
            Xs.append(0.5 * (M + M.T))
        cases.append((np.array(Xs), rng.standard_normal((3, N))))
    z = np.zeros((2, 40, 40))
    z[1, :10, :10] = cases[2][0][0][:10, :10]                     # all zero / rank 10: reflectors stop early
    cases.append((z, rng.standard_normal((2, 40))))
    for X, y in cases:
        B, N = X.shape[0], X.shape[1]
        Xs = np.array([x * 2.0**(1 - np.frexp(max(np.max(np.abs(x)), 1e-300))[1]) for x in X])
        dX, dy = ctx.to_device(Xs), ctx.to_device(y)
        dX1, dy1, dQ = ctx.empty((B, N, N)), ctx.empty((B, N)), ctx.empty((B, N, N))
        _lib.check(_lib.lib.vi_qr_similarity_f64(ctx.handle, B, N, dX.ptr, dy.ptr, dX1.ptr, dy1.ptr, dQ.ptr), 'qr')
        X1, y1, Q = dX1.download(), dy1.download(), dQ.download()
        for i in range(B):
            Qi = Q[i].T                                  # Q[:, j] contiguous
            Xl = np.tril(Xs[i]) + np.tril(Xs[i], -1).T
            sc = max(np.max(np.abs(Xl)), 1e-300)
            assert np.max(np.abs(Qi.T @ Qi - np.eye(N))) <= 1e-13
            assert np.max(np.abs(X1[i] - Qi.T @ Xl @ Qi)) <= 1e-13 * sc
            assert np.max(np.abs(X1[i] - X1[i].T)) <= 1e-13 * sc
            assert np.max(np.abs(y1[i] - Qi.T @ y[i])) <= 1e-13 * max(np.max(np.abs(y[i])), 1e-300)


def test_device_brent_equals_host_brent(monkeypatch):
    """vi_brent_warm_f64 (csrc/vi_brent.hip): Brent's iteration of a whole batch in one launch, a workgroup per record.  The
    same records through the host-driven iteration (VINTERP_DEVICE_BRENT=0) - rotated systems set up at the middle of the
    bracket, moved next to the root when the iterates cluster, moved again for the records that bisect a jump - give the same
    alpha, chi^2 and coefficients BIT FOR BIT, the same iteration counts and final brackets: the kernel runs the host path's
    arithmetic (the K3 body, the products of the re-basing, chi^2, 10^x) in the host path's order.  Also with the re-basing
    switched off on both sides."""
    from volumetricinterp_amd import synth
    m, ctx, eng, A, _ = _engine(CFG144, synth.GEOM_C2)
    P, T = A.shape[0], 24
    value, error = synth.synth_records(A, T, seed0=8000)
    W = error**-2.
    for rebase in ('1', '0'):
        monkeypatch.setenv('VINTERP_REBASE', rebase)
        monkeypatch.setenv('VINTERP_DEVICE_BRENT', '1')
        n0 = eng.stats.get('device_brent_records', 0)
        dev = eng.fit(W, value, [P] * T)
        ndev = eng.stats.get('device_brent_records', 0)
        assert ndev - n0 >= T // 2
        monkeypatch.setenv('VINTERP_DEVICE_BRENT', '0')
        host = eng.fit(W, value, [P] * T)
        assert eng.stats.get('device_brent_records', 0) == ndev            # none added: the host iterated
        monkeypatch.delenv('VINTERP_REBASE')
        monkeypatch.delenv('VINTERP_DEVICE_BRENT')
        for t in range(T):
            a1, a2 = dev['reg_params'][t]['curvature'], host['reg_params'][t]['curvature']
            i1, i2 = dev['search']['curvature']['info'][t], host['search']['curvature']['info'][t]
            assert i1.get('iterations') == i2.get('iterations') and i1.get('other_end') == i2.get('other_end'), (rebase, t, i1, i2)
            assert a1 == a2 or (np.isnan(a1) and np.isnan(a2)), (rebase, t, a1, a2)
            assert np.array_equal(dev['Coeffs'][t], host['Coeffs'][t], equal_nan=True), (rebase, t)
            assert dev['chi_sq'][t] == host['chi_sq'][t] or np.isnan(a1)
    eng.close()


def test_device_brent_with_more_than_16384_points_per_record(monkeypatch):
    """Records of 65 x 256 = 16 640 data points (65 blocks of chi^2 partial sums per record; until round 4 the kernel held 64 and
    the DEFAULT path of a batch of >= 8 records raised at this size - the reference accepts any P, interpolate.py:214).  Eight
    records through the default batch path (device-side Brent) and through the host-driven iteration: same bits."""
    from volumetricinterp_amd import _lib, synth
    geom = (65, 256)
    assert _lib.lib.vi_brent_warm_supported(144, geom[0] * geom[1]) == 1
    m, ctx, eng, A, _ = _engine(CFG144, geom)
    P, T = A.shape[0], 8
    assert P > 16384
    value, error = synth.synth_records(A, T, seed0=8100)
    W = error**-2.
    n0 = eng.stats.get('device_brent_records', 0)
    dev = eng.fit(W, value, [P] * T)                                   # no environment switch: what a user gets
    nroot = dev['search']['curvature']['outcomes'].count('root')
    assert nroot >= 1 and eng.stats.get('device_brent_records', 0) - n0 >= 1
    monkeypatch.setenv('VINTERP_DEVICE_BRENT', '0')
    host = eng.fit(W, value, [P] * T)
    monkeypatch.delenv('VINTERP_DEVICE_BRENT')
    for t in range(T):
        a1, a2 = dev['reg_params'][t]['curvature'], host['reg_params'][t]['curvature']
        assert a1 == a2 or (np.isnan(a1) and np.isnan(a2)), (t, a1, a2)
        assert np.array_equal(dev['Coeffs'][t], host['Coeffs'][t], equal_nan=True), t
    eng.close()


def test_role_separated_k3_gives_the_bits_of_the_two_barrier_kernel(tmp_path):
    """K3 exists twice (round 4): jacobi_system (two barriers per round, every order) and jacobi_system_v2 (vi_jacobi_v2_device.h:
    a set-up wave that keeps the diagonal blocks in registers and takes the cross parts from a mailbox, update waves with one
    block per thread; used from N = 93 on, in k_jacobi_solve and inside k_brent_warm).  Same ordering, same rotation formulas,
    same arithmetic per element: the eigenvalues and sweep counts of 24 systems (well-conditioned and graded, rank-deficient,
    indefinite) at N = 144 and N = 100, and a whole 12-record default-order fit (batch path: shared walk, device-side Brent,
    final solves, guard) must agree BIT FOR BIT between `VINTERP_K3=v1` and the default.  The choice is read once per process,
    hence two child processes."""
    import subprocess
    import sys
    script = tmp_path / 'k3_bits.py'
    script.write_text('''
import ctypes as C, io, os, sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
from volumetricinterp_amd import _lib, fitengine, synth
from test_gpu_configs import _engine, CFG144
out = {}
ctx = _lib.get_context()
_lib._sig("vi_eigvals_f64", C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP)
rng = np.random.default_rng(3)
for N in (144, 100):
    B = 12
    X = np.empty((B, N, N))
    for i in range(B):
        Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
        lam = rng.uniform(0.1, 1., N) if i %% 2 else 10.0**rng.uniform(-40, 0, N) * rng.choice([-1, 1], N)
        Mx = (Q * lam) @ Q.T
        X[i] = 0.5 * (Mx + Mx.T)
    dX, dl, ds = ctx.to_device(X), ctx.empty((B, N)), ctx.empty((B,), np.int32)
    _lib.check(_lib.lib.vi_eigvals_f64(ctx.handle, B, N, dX.ptr, dl.ptr, ds.ptr), "vi_eigvals_f64")
    out["lam%%d" %% N], out["sw%%d" %% N] = dl.download(), ds.download()
m, ctx, eng, A, _ = _engine(CFG144, synth.GEOM_C2)
P, T = A.shape[0], 12
value, error = synth.synth_records(A, T, seed0=1000)
res = eng.fit(error**-2., value, [P] * T)
out["C"], out["cov"], out["chi"] = res["Coeffs"], res["Covariance"], res["chi_sq"]
out["alpha"] = np.array([res["reg_params"][t]["curvature"] for t in range(T)])
out["its"] = np.array([res["search"]["curvature"]["info"][t].get("iterations", -1) or -1 for t in range(T)])
np.savez(sys.argv[1], **out)
''' % (REPO_ROOT, REPO_ROOT))
    got = {}
    for mode in ('v1', 'v2'):
        env = dict(os.environ)
        env.pop('VINTERP_K3', None)
        if mode == 'v1':
            env['VINTERP_K3'] = 'v1'
        o = str(tmp_path / ('out_%s.npz' % mode))
        r = subprocess.run([sys.executable, str(script), o], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        got[mode] = np.load(o)
    assert (got['v1']['sw144'] > 5).any()
    for k in got['v1'].files:
        assert np.array_equal(got['v1'][k], got['v2'][k], equal_nan=True), k
    assert np.isfinite(got['v2']['alpha']).sum() >= 8
