"""GPU parity: HIP basis / transform / fused-evaluation kernels (through the C-ABI) against the golden
vectors produced by the reference and against the CPU oracle on seeded inputs."""
import datetime as dt
import io
import warnings

import numpy as np
import pytest

from conftest import load_golden, colnorm_err, rel

pytestmark = pytest.mark.gpu

CFG = """[DEFAULT]
[MODEL]
NAME = {name}
MAXK = {k}
MAXL = {l}
CAP_LIM = {cap}
MAX_Z_INT = INF
LATCP = 78
LONCP = 262
EPS = 100000.0
LATRANGE = 74,80
LONRANGE = 260,285
ALTRANGE = 100,600
NUMGRIDPNT = {ngrid}
"""

SPH_VARIANTS = ['default', 'k8l2', 'k4l3', 'k3l4cap15', 'k2l5cap12p7', 'k2l3cap45', 'k8l12cap15', 'k2l12cap10']


def sph_model(cfg):
    from volumetricinterp_amd.models.sphharmlag import Model
    return Model(io.StringIO(CFG.format(name='sphharmlag', k=int(cfg[0]), l=int(cfg[1]), cap=repr(float(cfg[2])), ngrid=7)))


def rbf_model(ngrid=7):
    from volumetricinterp_amd.models.radbasfun import Model
    return Model(io.StringIO(CFG.format(name='radbasfun', k=4, l=6, cap=10, ngrid=ngrid)))


@pytest.mark.parametrize('tag', SPH_VARIANTS)
def test_transform_and_basis_vs_reference(tag):
    g = load_golden('basis_sph')
    m = sph_model(g[tag + '_cfg'])
    lat, lon, alt = g[tag + '_lat'], g[tag + '_lon'], g[tag + '_alt']
    z, t, p = m.transform_coord(lat, lon, alt)
    np.testing.assert_allclose(z, g[tag + '_z'], rtol=0, atol=1e-11)       # gate L1 (z spans 0..11, 1e-12 rel)
    np.testing.assert_allclose(t, g[tag + '_theta'], rtol=0, atol=1e-12)
    # phi is undefined at the rotated pole (theta ~ 6e-5 rad for point 1): compare where sin(theta) is not tiny
    ok = np.sin(g[tag + '_theta']) > 1e-3
    np.testing.assert_allclose(p[ok], g[tag + '_phi'][ok], rtol=0, atol=1e-12)
    A = m.basis(lat, lon, alt)
    Aref = g[tag + '_A']
    assert A.shape == Aref.shape
    assert np.array_equal(np.isnan(A), np.isnan(Aref))
    fin = np.isfinite(Aref).all(axis=0)
    # gate L2 on every point that is not next to the rotated pole ...
    err = colnorm_err(A[ok][:, fin], Aref[ok][:, fin])
    assert np.max(err) <= 1e-11, (tag, float(np.max(err)), int(np.argmax(err)))
    # ... where sin(theta) = sqrt(1 - x^2) is formed from x = 1 - 2e-9 and one ulp of x moves P_nu^m by
    # m * 3e-8 relative (measured with mpmath: SciPy itself is 1.7e-11 of the column maximum away from
    # the true value there at nu = 134.5); parity there is bounded by that conditioning, not by 1e-11.
    err_all = colnorm_err(A[:, fin], Aref[:, fin])
    assert np.max(err_all) <= 2e-9, (tag, float(np.max(err_all)), int(np.argmax(err_all)))


def test_basis_nd_and_layouts():
    g = load_golden('basis_sph')
    from volumetricinterp_amd import synth
    m = sph_model(g['default_cfg'])
    grid = synth.query_grid(3)
    A = m.basis(*grid)
    assert A.shape == (3, 3, 3, 144)
    assert np.max(colnorm_err(A, g['nd_A'])) <= 1e-11
    # N x P layout used by the fit kernels
    ctx = m.ctx
    P = grid[0].size
    d = [ctx.to_device(a.ravel()) for a in grid]
    At = m.basis_device(d[0], d[1], d[2], P, transposed=True).download()
    assert np.array_equal(At.T, A.reshape(P, 144))
    # empty input
    assert m.basis(np.zeros((0,)), np.zeros((0,)), np.zeros((0,))).shape == (0, 144)


def test_basis_vs_oracle_random_points():
    import oracle
    rng = np.random.default_rng(3)
    n = 3000                                   # more than one 256-thread block, ragged tail
    lat, lon, alt = rng.uniform(70, 86, n), rng.uniform(230, 290, n), rng.uniform(50e3, 900e3, n)
    for cfg in ([4, 6, 10.], [3, 4, 15.]):
        m = sph_model(cfg)
        o = oracle.SphHarmLagOracle(maxk=cfg[0], maxl=cfg[1], cap_lim_deg=cfg[2])
        A = m.basis(lat, lon, alt)
        Aref = o.basis(lat, lon, alt)
        assert np.max(colnorm_err(A, Aref)) <= 1e-11


def test_rbf_basis_vs_reference():
    g = load_golden('basis_rbf')
    m = rbf_model()
    np.testing.assert_allclose(m.centers, g['centers'], rtol=1e-15)
    A = m.basis(g['lat'], g['lon'], g['alt'])
    np.testing.assert_allclose(A, g['A'], rtol=2e-10, atol=1e-300)   # exp(-r^2/eps^2) with r^2/eps^2 up to ~1e3
    assert np.max(colnorm_err(A, g['A'])) <= 1e-11
    m3 = rbf_model(3)
    assert np.max(colnorm_err(m3.basis(g['lat'], g['lon'], g['alt']), g['g3_A'])) <= 1e-11
    from volumetricinterp_amd import synth
    assert m.basis(*synth.query_grid(2)).shape == (2, 2, 2, 343)
    R = m.transform_coords(g['lat'], g['lon'], g['alt'])
    import oracle
    np.testing.assert_allclose(R, np.array(oracle.geodetic2ecef(g['lat'], g['lon'], g['alt'])), rtol=1e-14)


def _estimate(tag):
    from volumetricinterp_amd.estimate import Estimate
    f = load_golden('fit_' + tag)
    return f, Estimate.from_arrays(f['Coeffs'], f['Covariance'], f['utime'], f['hull_vert'], str(f['cfg']))


@pytest.mark.parametrize('tag', ['k8l2', 'default'])
def test_estimate_vs_reference(tag):
    e = load_golden('eval')
    from volumetricinterp_amd import synth
    f, es = _estimate(tag)
    grid = synth.query_grid(6)
    t_mid = dt.datetime(1970, 1, 1) + dt.timedelta(seconds=float(e[tag + '_t_mid']))
    out = es(t_mid, *grid, check_hull=False)
    assert out.shape == (6, 6, 6)
    assert rel(out, e[tag + '_nohull']) <= 1e-10                         # gate L6
    outh = es(t_mid, *grid)                                              # check_hull=True is the default
    assert np.array_equal(np.isnan(outh), np.isnan(e[tag + '_hull']))
    ok = np.isfinite(outh)
    assert rel(outh[ok], e[tag + '_hull'][ok]) <= 1e-10
    assert np.array_equal(es.check_hull(*grid), np.isfinite(e[tag + '_hull']))
    es.timeinterp = True
    t_int = dt.datetime(1970, 1, 1) + dt.timedelta(seconds=float(e[tag + '_t_int']))
    assert rel(es(t_int, *grid, check_hull=False), e[tag + '_tinterp']) <= 1e-10
    es.timeinterp = False
    with pytest.raises(ValueError, match='Requested time out of range of data file.'):
        es(t_mid - dt.timedelta(seconds=4000), *grid)
    # calcgrad / calcerr are accepted and ignored
    assert np.array_equal(es(t_mid, *grid, calcgrad=True, calcerr=True, check_hull=False), out)


def test_eval_many_timesteps_and_ragged_sizes():
    """T = 22 rows (one 16-tile + one 4-tile + two single passes), Q not a multiple of the block size, vs the oracle."""
    import oracle
    f, es = _estimate('k8l2')
    rng = np.random.default_rng(8)
    Q = 777
    lat, lon, alt = rng.uniform(75, 81, Q), rng.uniform(250, 274, Q), rng.uniform(100e3, 700e3, Q)
    C = np.concatenate([f['Coeffs'], f['Coeffs'][:2] * 0.5])            # (6, 32)
    C = np.concatenate([C, -C, 3 * C, C[:4] + C[1:5]])                 # (22, 32)
    out = es.evaluate_coeffs(C, lat, lon, alt, check_hull=False)
    o = oracle.SphHarmLagOracle(maxk=8, maxl=2)
    A = o.basis(lat, lon, alt)
    assert C.shape[0] == 22
    for t in range(22):
        assert rel(out[t], A @ C[t]) <= 1e-10
    # hull mask vs the reference's per-point Qhull on a subset
    sub = slice(0, 120)
    outh = es.evaluate_coeffs(C[:1], lat[sub], lon[sub], alt[sub], check_hull=True)
    chk = oracle.check_hull(f['hull_vert'], lat[sub], lon[sub], alt[sub])
    assert np.array_equal(np.isfinite(outh[0]), chk)
    assert es.evaluate_coeffs(C, lat[:0], lon[:0], alt[:0]).shape == (22, 0)


@pytest.mark.parametrize('tag,maxk,maxl', [('default', 4, 6), ('k8l2', 8, 2)])
def test_eval_matrix_core_tiles_vs_oracle(tag, maxk, maxl):
    """T = 117 rows on one grid: 64 + 32 + 16 timesteps through the matrix-core kernel (k_eval_sph_mfma, tiles of
    4 / 2 / 1 x 16), the last 5 through the VALU kernels; Q = 1003 is not a multiple of 16 or 64; against the oracle's
    basis matrix, with the hull mask, and with the matrix-core path switched off."""
    import oracle
    f, es = _estimate(tag)
    rng = np.random.default_rng(21)
    Q, T = 1003, 117
    lat, lon, alt = rng.uniform(75, 81, Q), rng.uniform(250, 274, Q), rng.uniform(100e3, 700e3, Q)
    N = f['Coeffs'].shape[1]
    C = rng.standard_normal((T, N)) * np.abs(np.nan_to_num(f['Coeffs'][0])).max()
    out = es.evaluate_coeffs(C, lat, lon, alt, check_hull=False)
    A = oracle.SphHarmLagOracle(maxk=maxk, maxl=maxl).basis(lat, lon, alt)
    ref = C @ A.T
    assert out.shape == (T, Q)
    for t in range(T):
        assert rel(out[t], ref[t]) <= 1e-10, t
    outh = es.evaluate_coeffs(C, lat, lon, alt, check_hull=True)
    chk = oracle.check_hull(f['hull_vert'], lat[:150], lon[:150], alt[:150])
    assert np.array_equal(np.isfinite(outh[3, :150]), chk) and np.array_equal(np.isfinite(outh[116, :150]), chk)
    ok = np.isfinite(outh)
    assert np.array_equal(ok, np.broadcast_to(ok[0], ok.shape))
    assert rel(outh[ok], out[ok]) == 0.0                                # same kernel, masked stores only


@pytest.mark.parametrize('maxk,maxl,cap', [(4, 3, 10.), (2, 12, 15.), (8, 12, 15.), (4, 6, 12.7)])
def test_eval_matrix_core_other_orders_vs_oracle(maxk, maxl, cap):
    """Other orders through the multi-timestep dispatch: MAXL 3 x MAXK 4 (matrix-core kernel, integer degrees); MAXL 12 x
    MAXK 2 and the configs[4] order MAXL 12 x MAXK 8 at CAP_LIM 15 (half-integer degrees, 2F1 seeds; the high-order kernel
    k_eval_sph_split with the chains in two groups; at CAP_LIM 10 this
    order overflows Kvm, F8, and every density is NaN as in the reference); and CAP_LIM 12.7 (several degree groups: the
    generic kernel)."""
    import oracle
    from volumetricinterp_amd.estimate import Estimate
    from volumetricinterp_amd import synth
    cfg = ('[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = %d\nMAXL = %d\nCAP_LIM = %g\nMAX_Z_INT = INF\nLATCP = 78\n'
           'LONCP = 262\n' % (maxk, maxl, cap))
    N = maxk * maxl * maxl
    rng = np.random.default_rng(5)
    Q, T = 333, 40                                             # 32 + (8 through the VALU kernels)
    lat, lon, alt = rng.uniform(75, 81, Q), rng.uniform(250, 274, Q), rng.uniform(100e3, 700e3, Q)
    A = oracle.SphHarmLagOracle(maxk=maxk, maxl=maxl, cap_lim_deg=cap).basis(lat, lon, alt)
    good = np.all(np.isfinite(A), axis=0)
    assert np.all(good)
    C = rng.standard_normal((T, N)) / np.maximum(np.abs(A).max(axis=0), 1e-300)
    es = Estimate.from_arrays(C, None, synth.unix_times(T), np.zeros((4, 3)), cfg)
    out = es.evaluate_coeffs(C, lat, lon, alt, check_hull=False)
    ref = C[:, good] @ A[:, good].T
    for t in range(T):
        assert rel(out[t], ref[t]) <= 1e-10, t


def test_rbf_estimate_vs_oracle():
    import oracle
    from volumetricinterp_amd.estimate import Estimate
    f = load_golden('fit_rbf')
    es = Estimate.from_arrays(f['Coeffs'], f['Covariance'], f['utime'], f['hull_vert'], str(f['cfg']))
    rng = np.random.default_rng(9)
    Q = 300
    lat, lon, alt = rng.uniform(75, 81, Q), rng.uniform(255, 270, Q), rng.uniform(100e3, 600e3, Q)
    out = es.evaluate_coeffs(f['Coeffs'], lat, lon, alt, check_hull=False)
    o = oracle.RadBasFunOracle.from_config(io.StringIO(str(f['cfg'])))
    A = o.basis(lat, lon, alt)
    for t in range(f['Coeffs'].shape[0]):
        # |C| ~ 1e20 with heavy cancellation: compare against the size of the terms, not of the sum
        scale = np.linalg.norm(np.abs(A) @ np.abs(f['Coeffs'][t]))
        assert np.linalg.norm(out[t] - A @ f['Coeffs'][t]) <= 1e-11 * scale


def test_rbf_many_timesteps_tiles_agree():
    """The radial-basis evaluation computes each exponential once per tile of 16, 4 or 1 timesteps: 37 timesteps (two tiles
    of 16, one of 4, one single) against the oracle's basis, and row by row against the single-timestep launches."""
    import oracle
    from volumetricinterp_amd.estimate import Estimate
    f = load_golden('fit_rbf')
    rng = np.random.default_rng(10)
    T, N = 37, f['Coeffs'].shape[1]
    Cs = rng.standard_normal((T, N))
    es = Estimate.from_arrays(Cs, None, np.stack([np.arange(T), np.arange(T) + 1.], axis=1), f['hull_vert'], str(f['cfg']))
    Q = 1000
    lat, lon, alt = rng.uniform(75, 81, Q), rng.uniform(255, 270, Q), rng.uniform(100e3, 600e3, Q)
    out = es.evaluate_coeffs(Cs, lat, lon, alt, check_hull=False)
    A = oracle.RadBasFunOracle.from_config(io.StringIO(str(f['cfg']))).basis(lat, lon, alt)
    for t in range(T):
        assert rel(out[t], A @ Cs[t]) <= 1e-12, t
        one = es.evaluate_coeffs(Cs[t:t + 1], lat, lon, alt, check_hull=False)
        assert rel(out[t], one[0]) <= 1e-14, t


@pytest.mark.parametrize('tag', ['default', 'k3l4cap15', 'k2l5cap12p7'])
def test_grad_basis_vs_reference(tag):
    """Model.grad_basis (sphharmlag.py:148-184, next row N1): (P, 3, N) against the reference's own output."""
    g = load_golden('grad_sph')
    m = sph_model(g[tag + '_cfg'])
    G = m.grad_basis(g[tag + '_lat'], g[tag + '_lon'], g[tag + '_alt'])
    Gref = g[tag + '_G']
    assert G.shape == Gref.shape
    for c in range(3):
        err = colnorm_err(G[:, c, :], Gref[:, c, :])
        assert np.max(err) <= 1e-11, (tag, c, float(np.max(err)), int(np.argmax(err)))
    assert m.grad_basis(np.zeros(0), np.zeros(0), np.zeros(0)).shape == (0, 3, m.nbasis)


@pytest.mark.parametrize('tag', ['k8l2', 'default'])
def test_estimate_gradient_vs_oracle(tag):
    """Estimate.gradient (fused device contraction of grad_basis with the coefficients, vi_eval_grad_f64) against the
    oracle's grad_basis @ C, with the hull mask of the value path, on an N-D grid."""
    import oracle
    from volumetricinterp_amd import synth
    f, es = _estimate(tag)
    maxk, maxl = (8, 2) if tag == 'k8l2' else (4, 6)
    o = oracle.SphHarmLagOracle(maxk=maxk, maxl=maxl)
    grid = synth.query_grid(5)
    t_mid = dt.datetime(1970, 1, 1) + dt.timedelta(seconds=float(np.mean(f['utime'][0])))
    C, _ = oracle.get_C(t_mid, f['utime'], np.nan_to_num(f['Coeffs']), f['Covariance'])
    es.Coeffs = np.nan_to_num(es.Coeffs)
    g = es.gradient(t_mid, *grid, check_hull=False)
    assert g.shape == (5, 5, 5, 3)
    ref = oracle.evaluate_gradient(o, C, *grid)
    for c in range(3):
        assert rel(g[..., c], ref[..., c]) <= 1e-10, c
    gh = es.gradient(t_mid, *grid)                                       # hull mask on by default, like __call__
    refh = oracle.evaluate_gradient(o, C, *grid, hull_vert=f['hull_vert'])
    assert np.array_equal(np.isnan(gh), np.isnan(refh))
    ok = np.isfinite(refh)
    assert ok.any() and rel(gh[ok], refh[ok]) <= 1e-10
    # consistency with the materialised gradient basis
    G = es.model.grad_basis(grid[0].ravel(), grid[1].ravel(), grid[2].ravel())
    assert rel(g.reshape(-1, 3), np.einsum('pcn,n->pc', G, C)) <= 1e-12
    assert es.gradient(t_mid, grid[0][:0, 0, 0], grid[1][:0, 0, 0], grid[2][:0, 0, 0]).shape == (0, 3)


def test_estimate_error_vs_oracle():
    """Estimate.error = sqrt(a^T dC a) (basis tile -> rocBLAS GEMM with the covariance -> row dot, vi_eval_err_f64) against
    the oracle on the screened fixture, whose covariance is positive semidefinite to rounding; hull mask as __call__."""
    import oracle
    from volumetricinterp_amd import synth
    f, es = _estimate('k8l2')
    o = oracle.SphHarmLagOracle(maxk=8, maxl=2)
    grid = synth.query_grid(6)
    t_mid = dt.datetime(1970, 1, 1) + dt.timedelta(seconds=float(np.mean(f['utime'][0])))
    _, dC = oracle.get_C(t_mid, f['utime'], f['Coeffs'], f['Covariance'])
    e = es.error(t_mid, *grid, check_hull=False)
    ref = oracle.evaluate_error(o, dC, *grid)
    assert e.shape == (6, 6, 6)
    ok = np.isfinite(ref)
    assert ok.sum() > 100 and np.array_equal(np.isfinite(e), ok)
    assert rel(e[ok], ref[ok]) <= 1e-8
    eh = es.error(t_mid, *grid)
    refh = oracle.evaluate_error(o, dC, *grid, hull_vert=f['hull_vert'])
    assert np.array_equal(np.isnan(eh), np.isnan(refh))
    # many points: more than one 65 536-point chunk
    rng = np.random.default_rng(2)
    Q = 70001
    lat, lon, alt = rng.uniform(75, 81, Q), rng.uniform(250, 274, Q), rng.uniform(100e3, 700e3, Q)
    big = es.error(t_mid, lat, lon, alt, check_hull=False)
    sub = slice(65000, 65600)
    refb = oracle.evaluate_error(o, dC, lat[sub], lon[sub], alt[sub])
    okb = np.isfinite(refb)
    assert rel(big[sub][okb], refb[okb]) <= 1e-8
