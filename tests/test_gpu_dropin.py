"""Drop-in use of the reference's Python surface end to end on the GPU: config file + AMISR input file ->
Interpolate.calc_coeffs -> saveh5 -> Estimate(file)(time, lat, lon, alt), checked against the CPU oracle."""
import datetime as dt
import io
import os
import warnings

import numpy as np
import pytest

from conftest import load_golden, rel

pytestmark = pytest.mark.gpu

CFG = """[DEFAULT]
PARAM = dens
FILENAME = {inp}
OUTPUTFILENAME = {out}
REGULARIZATION_LIST = curvature
REGULARIZATION_METHOD = chi2
ERRLIM = 1e9,1e13
GOODFITCODE = 1,2,3,4
CHI2LIM = 0.1,10

[MODEL]
NAME = sphharmlag
MAXK = 8
MAXL = 2
CAP_LIM = 10
MAX_Z_INT = INF
LATCP = 78
LONCP = 262
"""


def test_config_to_hdf5_to_estimate(tmp_path):
    import oracle
    from volumetricinterp_amd import Interpolate, Estimate, synth, h5io
    f = load_golden('fit_k8l2')
    nb, nr = synth.GEOM_C1
    T = 3
    value, error = f['value'][:T].copy(), f['error'][:T].copy()
    value[1, 7] = 5e11                                   # the reader (not a NaN in the array) must drop this point
    inp, out = str(tmp_path / 'amisr.h5'), str(tmp_path / 'coeffs.h5')
    fitcode = np.ones((T, nb, nr), dtype=np.int64)
    fitcode[1].flat[7] = 9                               # bad fit code -> value/error NaN -> point dropped
    chi2 = np.ones((T, nb, nr))
    with h5io.H5File(inp, 'w') as h5:
        for g in ('/Time', '/Geomag', '/FittedParams', '/FittedParams/FitInfo'):
            h5.create_group(g)
        h5.create_array('/Time/UnixTime', f['utime'][:T])
        h5.create_array('/Geomag/Altitude', f['alt'].reshape(nb, nr))
        h5.create_array('/Geomag/Latitude', f['lat'].reshape(nb, nr))
        h5.create_array('/Geomag/Longitude', f['lon'].reshape(nb, nr))
        h5.create_array('/FittedParams/FitInfo/chi2', chi2)
        h5.create_array('/FittedParams/FitInfo/fitcode', fitcode)
        h5.create_array('/FittedParams/IonMass', np.array([16.]))
        h5.create_array('/FittedParams/Ne', value.reshape(T, nb, nr))
        h5.create_array('/FittedParams/dNe', error.reshape(T, nb, nr))
    cfgfile = str(tmp_path / 'config.ini')
    with open(cfgfile, 'w') as fh:
        fh.write(CFG.format(inp=inp, out=out))

    it = Interpolate(cfgfile)
    assert it.model.nbasis == 32 and it.regularization_list == ['curvature'] and it.reg_method == 'chi2'
    # the model's own regularisation matrix (host quadrature) equals the reference's
    np.testing.assert_array_equal(it.model.eval_reg_matricies['curvature'](), f['R'])
    it.calc_coeffs()
    assert it.Coeffs.shape == (T, 32) and it.Covariance.shape == (T, 32, 32) and it.chi_sq.shape == (T,)
    np.testing.assert_allclose(it.hull_vert, f['hull_vert'], rtol=1e-14)
    it.saveh5()
    assert os.path.exists(out)

    # oracle on the same inputs (value NaN where the reader masks)
    v_or = value.copy()
    v_or[1, 7] = np.nan
    o = oracle.SphHarmLagOracle(maxk=8, maxl=2)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        C, dC, c2, params = oracle.fit_records(o, f['lat'], f['lon'], f['alt'], v_or, error, {'curvature': f['R']},
                                               ['curvature'])
    for t in range(T):
        assert rel(it.Coeffs[t], C[t]) <= 1e-6
        assert rel(it.Covariance[t], dC[t]) <= 1e-5
        assert abs(it.chi_sq[t] - c2[t]) <= 1e-6 * c2[t]
    # records 0 and 2 are the reference's own golden records
    assert rel(it.Coeffs[0], f['Coeffs'][0]) <= 1e-6 and rel(it.Coeffs[2], f['Coeffs'][2]) <= 1e-6

    es = Estimate(out)
    assert es.model.nbasis == 32
    np.testing.assert_array_equal(es.Coeffs, it.Coeffs)
    g = synth.query_grid(5)
    t_mid = dt.datetime(1970, 1, 1) + dt.timedelta(seconds=float(np.mean(f['utime'][1])))
    dens = es(t_mid, *g)
    ref = oracle.evaluate(o, C[1], *g, hull_vert=f['hull_vert'])
    assert np.array_equal(np.isnan(dens), np.isnan(ref))
    ok = np.isfinite(ref)
    assert ok.sum() > 10 and rel(dens[ok], ref[ok]) <= 1e-6
    # time window selection (interpolate.py:503-507)
    it2 = Interpolate(cfgfile)
    it2.calc_coeffs(starttime=dt.datetime(1970, 1, 1) + dt.timedelta(seconds=float(f['utime'][1, 0])),
                    endtime=dt.datetime(1970, 1, 1) + dt.timedelta(seconds=float(f['utime'][2, 1])))
    assert it2.Coeffs.shape == (2, 32)
    # the same record in a batch of 2 instead of 3: a record's answer does not depend on the batch it is fitted in (the
    # multisection's sample count is independent of T since round 3), bit for bit
    assert np.array_equal(it2.Coeffs[1], it.Coeffs[2])


def _write_amisr(path, f, T):
    from volumetricinterp_amd import synth, h5io
    nb, nr = synth.GEOM_C1
    with h5io.H5File(path, 'w') as h5:
        for g in ('/Time', '/Geomag', '/FittedParams', '/FittedParams/FitInfo'):
            h5.create_group(g)
        h5.create_array('/Time/UnixTime', f['utime'][:T])
        h5.create_array('/Geomag/Altitude', f['alt'].reshape(nb, nr))
        h5.create_array('/Geomag/Latitude', f['lat'].reshape(nb, nr))
        h5.create_array('/Geomag/Longitude', f['lon'].reshape(nb, nr))
        h5.create_array('/FittedParams/FitInfo/chi2', np.ones((T, nb, nr)))
        h5.create_array('/FittedParams/FitInfo/fitcode', np.ones((T, nb, nr), dtype=np.int64))
        h5.create_array('/FittedParams/IonMass', np.array([16.]))
        h5.create_array('/FittedParams/Ne', f['value'][:T].reshape(T, nb, nr))
        h5.create_array('/FittedParams/dNe', f['error'][:T].reshape(T, nb, nr))


def test_cli_two_ranks_equal_one(tmp_path):
    """The command line entry point under a 2-rank launch (both ranks share this box's one GPU; the RCCL
    communicator refuses the duplicate device and the shared parameters go over the control socket) writes the same
    coefficient file as the single-process run."""
    import subprocess
    import sys
    from volumetricinterp_amd.estimate import Estimate
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f = load_golden('fit_k8l2')
    T = 4
    inp = str(tmp_path / 'amisr.h5')
    _write_amisr(inp, f, T)
    outs = []
    for world in (1, 2):
        out = str(tmp_path / ('coeffs%d.h5' % world))
        cfg = str(tmp_path / ('config%d.ini' % world))
        with open(cfg, 'w') as fh:
            fh.write(CFG.format(inp=inp, out=out))
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT='29517', PYTHONPATH=REPO, VINTERP_RDV_PATH=str(tmp_path / ('rdv%d.sock' % world)))
            procs.append(subprocess.Popen([sys.executable, '-m', 'volumetricinterp_amd.run_volumetricinterp', cfg],
                                          env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        for p_ in procs:
            o, _ = p_.communicate(timeout=600)
            assert p_.returncode == 0, o.decode()
        outs.append(Estimate(out))
    one, two = outs
    assert one.Coeffs.shape == two.Coeffs.shape == (T, 32)
    np.testing.assert_array_equal(one.time, two.time)
    for t in range(T):
        # 2 records per rank instead of 4 in one process: the same bits (records do not see each other)
        assert np.array_equal(two.Coeffs[t], one.Coeffs[t]), t
        assert rel(two.Coeffs[t], f['Coeffs'][t]) <= 1e-6, t


def test_unsupported_regularisation_name_raises_keyerror(tmp_path):
    from volumetricinterp_amd import Interpolate
    cfgfile = str(tmp_path / 'config.ini')
    with open(cfgfile, 'w') as fh:
        fh.write(CFG.format(inp='x.h5', out='y.h5').replace('REGULARIZATION_LIST = curvature',
                                                             'REGULARIZATION_LIST = bogus'))
    it = Interpolate(cfgfile)
    with pytest.raises(KeyError):                        # interpolate.py:488-493
        it.calc_coeffs()
