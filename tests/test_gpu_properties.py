"""Size-independent properties of the GPU path at BASELINE's full sizes (where the CPU oracle would take
minutes to hours): linearity of the evaluation in the coefficients, agreement of the kernel variants and
timestep tiles, invariance of a record's fit to the batch it is fitted in (records are independent)."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden, rel

pytestmark = pytest.mark.gpu

CFG = ('[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _eval(C, grid_n=128, hull=None):
    from volumetricinterp_amd import synth
    from volumetricinterp_amd.estimate import Estimate
    es = Estimate.from_arrays(C, None, synth.unix_times(len(C)), hull if hull is not None else np.zeros((4, 3)), CFG)
    g = synth.query_grid(grid_n)
    return es.evaluate_coeffs(C, *g, check_hull=hull is not None)


def test_eval_128cubed_linearity_and_tiles():
    """configs[1] grid (128^3 = 2 097 152 points): eval(a C1 + b C2) = a eval(C1) + b eval(C2); a row gives the
    same values whether it is evaluated alone (tile of 1), in a tile of 4 or in a tile of 16."""
    rng = np.random.default_rng(0)
    C = rng.standard_normal((21, 144))
    C[2] = 2.5 * C[0] - 0.75 * C[1]
    out = _eval(C)                                     # tiles 16 + 4 + 1
    assert out.shape == (21, 128**3) and np.all(np.isfinite(out))
    assert rel(out[2], 2.5 * out[0] - 0.75 * out[1]) <= 1e-12
    alone = _eval(C[20:21])                            # same row through the 1-tile kernel
    assert rel(alone[0], out[20]) <= 1e-13
    four = _eval(C[16:20])
    assert rel(four[3], out[19]) <= 1e-13
    sixteen = _eval(np.concatenate([C[5:21]]))         # rows 5..20 in a 16-tile
    assert rel(sixteen[0], out[5]) <= 1e-13 and rel(sixteen[15], out[20]) <= 1e-13


def test_fast_and_generic_eval_kernels_agree_at_128cubed():
    code = ("import sys, numpy as np\nsys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_gpu_properties import _eval\n"
            "C = np.random.default_rng(1).standard_normal((2, 144))\n"
            "np.save(sys.argv[1], _eval(C))\n") % (REPO, os.path.join(REPO, 'tests'))
    outs = []
    for mode in ('fast', 'generic'):
        fn = '/tmp/vi_prop_%s.npy' % mode
        env = dict(os.environ, VINTERP_EVAL=mode)
        r = subprocess.run([sys.executable, '-c', code, fn], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(np.load(fn))
        os.remove(fn)
    assert rel(outs[0], outs[1]) <= 1e-12


def test_hull_mask_is_independent_of_the_coefficients():
    f = load_golden('fit_default')
    rng = np.random.default_rng(2)
    C = rng.standard_normal((5, 144))
    out = _eval(C, grid_n=64, hull=f['hull_vert'])
    mask = np.isnan(out)
    assert 0.1 < mask[0].mean() < 0.9
    assert np.all(mask == mask[0])


def test_hull_mask_pass_on_surface_points_ragged_sizes_and_nonfinite_coordinates():
    """The mask pass (k_hull_mask_mx: plane distances as fp16-split products on the matrix cores, exact fp64 test of the points
    within the band of the surface spread over the lanes of a wave) against the definition in NumPy fp64 -
    inside <=> max_f (n_f . x + d_f) <= tol (estimate.py:153-178) - on points chosen to sit ON the hull (its own vertices,
    which the fp32 pass cannot decide), next to it (geodetic blends of neighbouring vertices), on query sizes that are not
    multiples of the 1024 points of a workgroup, on non-finite coordinates (outside), on points and facet lists outside the
    prefilter's range."""
    from scipy.spatial import ConvexHull
    from volumetricinterp_amd import synth
    from volumetricinterp_amd.estimate import Estimate, hull_equations
    from volumetricinterp_amd.geodesy import geodetic2ecef
    lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
    R = np.array(geodetic2ecef(lat, lon, alt)).T
    vid = ConvexHull(R).vertices
    hv = R[vid]
    eq, tol = hull_equations(hv)
    es = Estimate.from_arrays(np.zeros((1, 144)), None, synth.unix_times(1), hv, CFG)
    rng = np.random.default_rng(5)
    # blends of pairs of hull vertices in geodetic coordinates: a few metres to kilometres off the surface, both sides
    i, j = rng.integers(0, len(vid), 3000), rng.integers(0, len(vid), 3000)
    w = rng.random(3000)
    bl = [w * a[vid][i] + (1 - w) * a[vid][j] for a in (lat, lon, alt)]
    g = synth.query_grid(16)
    qlat = np.concatenate([lat, bl[0], g[0].ravel()])
    qlon = np.concatenate([lon, bl[1], g[1].ravel()])
    qalt = np.concatenate([alt, bl[2], g[2].ravel()])
    X = np.array(geodetic2ecef(qlat, qlon, qalt)).T
    d = (X @ eq[:, :3].T + eq[:, 3]).max(axis=1)
    expect = d <= tol
    decided = np.abs(d - tol) > 1e-6                # the device's ECEF differs from NumPy's in the last bits (~1e-9 m)
    assert (~decided).sum() < 400 and 200 < expect.sum() < len(expect) - 200
    for Q in (len(qlat), 1, 63, 1025, 2049, 5000):
        got = es.check_hull(qlat[:Q], qlon[:Q], qalt[:Q])
        assert np.array_equal(got[decided[:Q]], expect[:Q][decided[:Q]]), Q
    # the hull's own vertices are inside (Qhull's coplanar points, estimate.py:174-176) - all of them in the fp64 band
    on = es.check_hull(lat[vid], lon[vid], alt[vid])
    dv = (hv @ eq[:, :3].T + eq[:, 3]).max(axis=1)
    assert np.all(on[dv <= tol - 1e-9]) and on.mean() > 0.9
    # non-finite coordinates: outside, and they do not disturb their neighbours in the wave
    bad = qlat[:300].copy(), qlon[:300].copy(), qalt[:300].copy()
    bad[0][5] = np.nan
    bad[1][70] = np.inf
    bad[2][131] = -np.inf
    got = es.check_hull(*bad)
    ref = es.check_hull(qlat[:300], qlon[:300], qalt[:300])
    assert not got[5] and not got[70] and not got[131]
    keep = np.ones(300, dtype=bool)
    keep[[5, 70, 131]] = False
    assert np.array_equal(got[keep], ref[keep])
    # points the fp16 planes of the matrix-core pass do not reach (further than 16 000 km from its reference point: the far side
    # of the Earth, 20 000 km up) and angles beyond the range of its own sine / cosine (longitude + 2000 turns): decided by the
    # fp64 test, same answers, neighbours undisturbed
    far = qlat[:300].copy(), qlon[:300].copy(), qalt[:300].copy()
    far[0][7], far[1][7], far[2][7] = -75., 80., 9.0e5
    far[2][40] = 2.0e7
    far[1][100:164] = far[1][100:164] + 360. * 2000.
    got = es.check_hull(*far)
    assert not got[7] and not got[40]
    keep = np.ones(300, dtype=bool)
    keep[[7, 40]] = False
    keep[100:164] = np.abs(d[100:164] - tol) > 1e-3               # (the shifted longitudes move a point by ~1e-5 m)
    assert np.array_equal(got[keep], ref[keep])
    # the facet list as the C-ABI takes it: normals that are not unit vectors (every equation and the tolerance times 3, then
    # times 1e-3: the pass scales by the longest normal), a list with a non-finite entry (the definition puts every point
    # outside: NaN <= tol is false), offsets beyond the fp16 range (every point through the fp64 test)
    import ctypes as C
    from volumetricinterp_amd import _lib
    m, ctx = es.model, es.model.ctx
    Q = len(qlat)
    dq = [ctx.to_device(np.ascontiguousarray(a)) for a in (qlat, qlon, qalt)]
    dC, dout = ctx.to_device(np.zeros((1, m.nbasis))), ctx.empty((1, Q))

    def mask_of(e, t):
        de = ctx.to_device(np.ascontiguousarray(e))
        _lib.check(_lib.lib.vi_eval_f64(m.handle(), Q, dq[0].ptr, dq[1].ptr, dq[2].ptr, 1, dC.ptr, de.ptr, len(e), float(t), dout.ptr),
                   'vi_eval_f64')
        return np.isfinite(dout.download()[0])
    base = mask_of(eq, tol)
    assert np.array_equal(base[decided], expect[decided])
    for f in (3., 1e-3):
        assert np.array_equal(mask_of(eq * f, tol * f)[decided], expect[decided]), f
    bad_eq = eq.copy()
    bad_eq[17, 1] = np.nan
    assert not mask_of(bad_eq, tol).any()
    shifted = eq.copy()
    shifted[:, 3] -= 1e8
    assert mask_of(shifted, tol).all()
    shifted[:, 3] += 2e8
    assert not mask_of(shifted, tol).any()


def test_packed_fp32_hull_pass_gives_the_same_mask(tmp_path):
    """VINTERP_HULL=fp32 (read once per process: a subprocess) runs the packed-fp32 plane loop of rounds 2-3 (k_hull_mask)
    instead of the matrix-core pass; both hand their band to the same fp64 test, so the masks are the same bit for bit."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from scipy.spatial import ConvexHull\n"
        "from volumetricinterp_amd import synth\n"
        "from volumetricinterp_amd.estimate import Estimate\n"
        "from volumetricinterp_amd.geodesy import geodetic2ecef\n"
        "lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)\n"
        "R = np.array(geodetic2ecef(lat, lon, alt)).T\n"
        "es = Estimate.from_arrays(np.zeros((1, 144)), None, synth.unix_times(1), R[ConvexHull(R).vertices], %r)\n"
        "g = synth.query_grid(40)\n"
        "np.save(sys.argv[1], es.check_hull(*g))\n" % (root, CFG))
    masks = []
    for mode in ('', 'fp32'):
        env = dict(os.environ)
        env.pop('VINTERP_HULL', None)
        if mode:
            env['VINTERP_HULL'] = mode
        out = str(tmp_path / ('mask_%s.npy' % (mode or 'mx')))
        subprocess.run([sys.executable, '-c', script, out], check=True, env=env, timeout=300)
        masks.append(np.load(out))
    assert 0.2 < masks[0].mean() < 0.8
    assert np.array_equal(masks[0], masks[1])


def test_record_fit_is_independent_of_its_batch(tmp_path):
    """Records carry no state into each other (interpolate.py:511): fitting a record alone or inside a batch of
    others gives the same coefficients (screened, well-conditioned model; 1e-9 covers the different GEMM shapes)."""
    from test_gpu_fit import make_interp, reg_of
    f = load_golden('fit_k8l2')
    regm, reg = reg_of(f)
    it = make_interp(tmp_path, str(f['cfg']))
    full = it.fit_records(f['lat'], f['lon'], f['alt'], f['value'], f['error'], regm)
    for t in (0, 3):
        one = it.fit_records(f['lat'], f['lon'], f['alt'], f['value'][t:t + 1], f['error'][t:t + 1], regm)
        assert one['reg_params'][0][reg] == pytest.approx(full['reg_params'][t][reg], rel=1e-9)
        assert rel(one['Coeffs'][0], full['Coeffs'][t]) <= 1e-9
        assert rel(one['Covariance'][0], full['Covariance'][t]) <= 1e-8
    # reversed batch order: same rows, reversed
    rev = it.fit_records(f['lat'], f['lon'], f['alt'], f['value'][::-1], f['error'][::-1], regm)
    for t in range(4):
        assert rel(rev['Coeffs'][3 - t], full['Coeffs'][t]) <= 1e-9
