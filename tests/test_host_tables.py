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
