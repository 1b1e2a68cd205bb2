"""GPU parity of the resident-basis evaluation (vi_eval_basis_f64 + vi_eval_resident_f64: many timesteps on one grid as one
matrix product) against the reference's golden evaluation vectors, the CPU oracle and the fused kernel (vi_eval_f64)."""
import datetime as dt

import numpy as np
import pytest

from conftest import load_golden, rel

pytestmark = pytest.mark.gpu


def _estimate(tag):
    from volumetricinterp_amd.estimate import Estimate
    f = load_golden('fit_' + tag)
    return f, Estimate.from_arrays(f['Coeffs'], f['Covariance'], f['utime'], f['hull_vert'], str(f['cfg']))


@pytest.mark.parametrize('tag', ['k8l2', 'default'])
def test_resident_grid_vs_reference(tag):
    """The reference's own evaluation vectors (estimate.py:110-123 on the 6^3 grid of tools/gen_golden.py): hull mask identical,
    values to the 1e-10 of gate L6; Estimate.get_C picks the timestep as in __call__."""
    e = load_golden('eval')
    from volumetricinterp_amd import synth
    f, es = _estimate(tag)
    grid = synth.query_grid(6)
    t_mid = dt.datetime(1970, 1, 1) + dt.timedelta(seconds=float(e[tag + '_t_mid']))
    g0 = es.resident_grid(*grid, check_hull=False)
    out = g0([t_mid])
    assert out.shape == (1, 6, 6, 6)
    assert rel(out[0], e[tag + '_nohull']) <= 1e-10
    g1 = es.resident_grid(*grid)                                         # check_hull=True is the default, as in __call__
    outh = g1([t_mid, t_mid])
    for k in range(2):
        assert np.array_equal(np.isnan(outh[k]), np.isnan(e[tag + '_hull']))
        ok = np.isfinite(outh[k])
        assert rel(outh[k][ok], e[tag + '_hull'][ok]) <= 1e-10
    with pytest.raises(ValueError, match='Requested time out of range of data file.'):
        g1([t_mid - dt.timedelta(seconds=4000)])
    g0.close()
    g1.close()


@pytest.mark.parametrize('Q', [1777, 2052, 8192 + 256])
def test_resident_grid_many_timesteps_vs_oracle_and_fused_kernel(Q):
    """300 timesteps (four full tiles of 64 and a rest of 44) on random points, some outside the hull, one timestep with NaN
    coefficients (a failed fit): the oracle's A @ C to 1e-10, the fused kernel's values to 1e-12, its NaNs exactly.
    Q = 1777 (odd) goes through the library's product, the others through K2r - 2052 with a ragged last group of points."""
    import oracle
    f, es = _estimate('k8l2')
    rng = np.random.default_rng(18)
    lat, lon, alt = rng.uniform(75, 81, Q), rng.uniform(250, 274, Q), rng.uniform(100e3, 700e3, Q)
    base = np.concatenate([f['Coeffs'], f['Coeffs'][:2] * 0.5])          # (6, 32)
    C = np.concatenate([base * s for s in rng.uniform(-2., 2., 50)])     # (300, 32)
    C[123] = np.nan
    g = es.resident_grid(lat, lon, alt, check_hull=True)
    out = g.evaluate_coeffs(C)
    fused = es.evaluate_coeffs(C, lat, lon, alt, check_hull=True)
    assert out.shape == fused.shape == (300, Q)
    assert np.array_equal(np.isnan(out), np.isnan(fused))
    assert np.all(np.isnan(out[123]))
    inside = np.isfinite(fused[0])
    assert 0 < inside.sum() < Q
    o = oracle.SphHarmLagOracle(maxk=8, maxl=2)
    A = o.basis(lat, lon, alt)
    for t in (0, 1, 127, 128, 255, 256, 299):
        assert rel(out[t][inside], (A @ C[t])[inside]) <= 1e-10
        assert rel(out[t][inside], fused[t][inside]) <= 1e-12
    chk = oracle.check_hull(f['hull_vert'], lat[:150], lon[:150], alt[:150])
    assert np.array_equal(np.isfinite(out[0][:150]), chk)
    assert g.evaluate_coeffs(C[:0]).shape == (0, Q)
    g.close()
    empty = es.resident_grid(lat[:0], lon[:0], alt[:0])
    assert empty.evaluate_coeffs(C[:3]).shape == (3, 0)


def test_resident_default_order_both_paths(monkeypatch):
    """Default order (N = 144, 36 k-steps) on 4096 points x 70 timesteps: K2r against the library's product of the same basis
    (1e-13) and against the fused kernel (1e-12)."""
    f, es = _estimate('default')
    rng = np.random.default_rng(5)
    Q = 4096
    lat, lon, alt = rng.uniform(75, 81, Q), rng.uniform(250, 274, Q), rng.uniform(100e3, 700e3, Q)
    C = np.concatenate([f['Coeffs'] * s for s in rng.uniform(-2., 2., 35)])[:70]
    assert C.shape == (70, 144)
    g = es.resident_grid(lat, lon, alt, check_hull=True)
    own = g.evaluate_coeffs(C)
    fused = es.evaluate_coeffs(C, lat, lon, alt, check_hull=True)
    from volumetricinterp_amd import _lib
    ctx = es.model.ctx
    dC, dO = ctx.to_device(C), ctx.empty((70, Q + 1))
    # the library path: the same call with an output whose rows are not 32-byte aligned
    _lib.check(_lib.lib.vi_eval_resident_f64(es.model.handle(), Q, 70, g.dY.ptr, dC.ptr, dO.offset_ptr(1)), 'vi_eval_resident_f64')
    lib = dO.download().ravel()[1:1 + 70 * Q].reshape(70, Q)
    assert np.array_equal(np.isnan(own), np.isnan(fused)) and np.array_equal(np.isnan(own), np.isnan(lib))
    ok = np.isfinite(fused)
    assert 0 < ok[0].sum() < Q
    assert rel(own[ok], lib[ok]) <= 1e-13
    assert rel(own[ok], fused[ok]) <= 1e-12
    g.close()
