"""CPU checks of the host-side model tables (Model.device_tables) through a NumPy emulation of the
device recurrence, against the reference's golden basis matrices (gate L2: 1e-11 per column)."""
import io
import os
import sys

import numpy as np
import pytest

from conftest import load_golden, colnorm_err

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from emulate import basis_from_tables  # noqa: E402

CFG = """[DEFAULT]
[MODEL]
NAME = sphharmlag
MAXK = {k}
MAXL = {l}
CAP_LIM = {cap}
MAX_Z_INT = INF
LATCP = 78
LONCP = 262
"""


def make_model(cfg):
    from volumetricinterp_amd.models.sphharmlag import Model
    return Model(io.StringIO(CFG.format(k=int(cfg[0]), l=int(cfg[1]), cap=repr(float(cfg[2])))))


@pytest.mark.parametrize('tag', ['default', 'k8l2', 'k4l3', 'k3l4cap15', 'k2l5cap12p7', 'k2l3cap45', 'k8l12cap15',
                                 'k2l12cap10'])
def test_tables_reproduce_reference_basis(tag):
    g = load_golden('basis_sph')
    m = make_model(g[tag + '_cfg'])
    tb = m.device_tables()
    A = basis_from_tables(tb, m.maxk, m.maxl, g[tag + '_lat'], g[tag + '_lon'], g[tag + '_alt'])
    Aref = g[tag + '_A']
    assert np.array_equal(np.isnan(A), np.isnan(Aref))
    fin = np.isfinite(Aref).all(axis=0)
    err = colnorm_err(A[:, fin], Aref[:, fin])
    assert np.max(err) <= 1e-11, (tag, float(np.max(err)), int(np.argmax(err)))
    np.testing.assert_array_equal(np.array([m.nu(n) for n in range(m.nbasis)]), g[tag + '_nu'])


def test_default_groups():
    g = load_golden('basis_sph')
    m = make_model(g['default_cfg'])
    tb = m.device_tables()
    # nu = 4, 22, 40, 58.00000000000001, 76, 94 -> one recurrence (fractional parts within SNAP_TOL)
    assert len(tb['groups']) == 1 and tb['groups'][0]['nvmax'] == 94 and tb['groups'][0]['nterms'] == 0
    assert [int(j) for j in np.nonzero(tb['groups'][0]['pick'] >= 0)[0]] == [4, 22, 40, 58, 76, 94]


def test_psi_by_gauss_quadrature_matches_the_reference():
    """SURVEY 8f row N4, the part that is well defined: Psi (sphharmlag.py:215-239) from an exact Gauss-Laguerre z-integral,
    a Gauss-Legendre theta-integral and the analytic phi-integral (regmat.eval_psi_gauss) against the reference's own Psi
    (tests/golden/regmat.npz, scipy.integrate.quad with its default 1.49e-8 tolerances): 5e-10 of max|Psi| on three
    orders, and converged in itself to 1e-13 (so the difference is QUADPACK's).  Omega is not attempted: its z-integral
    diverges and the reference's values are QUADPACK artefacts."""
    import io
    from conftest import load_golden
    from volumetricinterp_amd import regmat
    from volumetricinterp_amd.models.sphharmlag import Model
    ref = load_golden('regmat')
    for tag, (k, l) in (('default', (4, 6)), ('k8l2', (8, 2)), ('k4l3', (4, 3))):
        cfg = ('[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = %d\nMAXL = %d\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\n'
               'LONCP = 262\n' % (k, l))
        m = Model(io.StringIO(cfg))
        P = regmat.eval_psi_gauss(m)
        R = ref[tag + '_0thorder']
        assert P.shape == R.shape and np.array_equal(P, P.T)
        assert np.max(np.abs(P - R)) <= 5e-10 * np.max(np.abs(R)), tag
        P2 = regmat.eval_psi_gauss(m, ntheta=1200)
        assert np.max(np.abs(P - P2)) <= 1e-13 * np.max(np.abs(P))



def test_default_psi_is_the_reference_s_and_finite_limits_converge(monkeypatch):
    """What the plug-in hands to the fit by default is the reference's own Psi BIT FOR BIT (regmat.eval_psi: its integrands and
    quad calls); the Gauss-quadrature variant is opt-in (VINTERP_REGMAT=gauss).  With a FINITE MAX_Z_INT - small, or 1e3 used
    as 'practically infinite' (a fixed Gauss-Legendre rule was unconverged there) - the opt-in variant agrees
    with the QUADPACK restatement to 5e-10 of max|Psi| as well."""
    import io
    from conftest import load_golden
    from volumetricinterp_amd import regmat
    from volumetricinterp_amd.models.sphharmlag import Model
    ref = load_golden('regmat')
    cfg = ('[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = %d\nMAXL = %d\nCAP_LIM = 10\nMAX_Z_INT = %s\nLATCP = 78\n'
           'LONCP = 262\n')
    monkeypatch.delenv('VINTERP_REGMAT', raising=False)
    m = Model(io.StringIO(cfg % (8, 2, 'INF')))
    assert np.array_equal(m.eval_reg_matricies['0thorder'](), ref['k8l2_0thorder'])
    monkeypatch.setenv('VINTERP_REGMAT', 'gauss')
    assert np.array_equal(m.eval_reg_matricies['0thorder'](), regmat.eval_psi_gauss(m))
    for zmax in ('3', '40', '1000'):
        mf = Model(io.StringIO(cfg % (4, 3, zmax)))
        Pq, Pg = regmat.eval_psi(mf), regmat.eval_psi_gauss(mf)
        assert np.max(np.abs(Pg - Pq)) <= 5e-10 * np.max(np.abs(Pq)), zmax


def test_regmat_worker_processes_give_the_same_bits(monkeypatch):
    """The angular integrals of Omega / Psi spread over worker processes (regmat._tp_parallel: from 3000 distinct (l, m) pairs
    on, i.e. at the doubled order of BASELINE configs[4], where one process needs 65 s for Omega): the same QUADPACK calls on
    the same scalars - the matrices are the reference's BIT FOR BIT whichever process computed an entry."""
    import io
    from conftest import load_golden
    from volumetricinterp_amd import regmat
    from volumetricinterp_amd.models.sphharmlag import Model
    ref = load_golden('regmat')
    m = Model(io.StringIO('[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 3\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\n'
                          'LONCP = 262\n'))
    monkeypatch.setenv('VINTERP_REGMAT_WORKERS', '2')
    assert np.array_equal(regmat.eval_omega(m), ref['k4l3_curvature'])
    assert np.array_equal(regmat.eval_psi(m), ref['k4l3_0thorder'])
    monkeypatch.delenv('VINTERP_REGMAT_WORKERS')
    assert regmat._workers(1296) == 1 and regmat._workers(20736) >= 1
