"""HDF5 coefficient-file layout and the AMISR input reader, through the ctypes libhdf5 shim."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import load_golden

h5io = pytest.importorskip('volumetricinterp_amd.h5io')
try:
    h5io._load()
except h5io.H5Error as e:                                   # pragma: no cover
    pytest.skip('libhdf5 not available: %s' % e, allow_module_level=True)


def test_coeff_file_roundtrip(tmp_path):
    f = load_golden('fit_k8l2')
    fn = str(tmp_path / 'coeffs.h5')
    cfg = str(f['cfg'])
    h5io.write_coeff_file(fn, time=f['utime'], Coeffs=f['Coeffs'], Covariance=f['Covariance'], reglist=['curvature'],
                          regmethod='chi2', chi2=f['chi_sq'], hull_vert=f['hull_vert'], rawfilename='synthetic.h5',
                          config_name='config.ini', config_path='/some/where', config_contents=cfg)
    d = h5io.read_coeff_file(fn)
    np.testing.assert_array_equal(d['Coeffs'], f['Coeffs'])
    np.testing.assert_array_equal(d['Covariance'], f['Covariance'])
    np.testing.assert_array_equal(d['time'], f['utime'])
    np.testing.assert_array_equal(d['hull_vert'], f['hull_vert'])
    assert d['config_file_text'].decode('utf-8') == cfg           # estimate.py:41 decodes the bytes
    with h5io.H5File(fn) as h5:
        assert h5.read('/FitParams/reglist').tolist() == [b'curvature']
        assert h5.read('/FitParams/regmethod') == b'chi2'
        assert h5.read('/RawData/filename') == b'synthetic.h5'
        assert h5.read('/ConfigFile/Name') == b'config.ini'
        np.testing.assert_array_equal(h5.read('/FitParams/chi2'), f['chi_sq'])
        with pytest.raises(h5io.H5Error):
            h5.read('/Coeffs/nope')
    with pytest.raises(h5io.H5Error):
        h5io.read_coeff_file(str(tmp_path / 'missing.h5'))


def test_empty_reglist(tmp_path):
    """radbasfun: REGULARIZATION_LIST is empty -> zero-length array node."""
    fn = str(tmp_path / 'c.h5')
    h5io.write_coeff_file(fn, time=np.zeros((1, 2)), Coeffs=np.zeros((1, 3)), Covariance=np.zeros((1, 3, 3)),
                          reglist=[], regmethod='chi2', chi2=np.zeros(1), hull_vert=np.zeros((4, 3)),
                          rawfilename='x', config_name='c', config_path='p', config_contents='[MODEL]\n')
    with h5io.H5File(fn) as h5:
        assert h5.read('/FitParams/reglist').shape == (0,)


@pytest.mark.skipif(not os.path.exists('/opt/conda/bin/python3.9'), reason='no independent HDF5 reader in this image')
def test_file_is_readable_by_an_independent_hdf5_library(tmp_path):
    """h5py (in the conda python of this image) sees the PyTables-style layout: fixed-length strings,
    CLASS/VERSION/TITLE attributes, the dataset shapes."""
    f = load_golden('fit_k8l2')
    fn = str(tmp_path / 'coeffs.h5')
    h5io.write_coeff_file(fn, time=f['utime'], Coeffs=f['Coeffs'], Covariance=f['Covariance'], reglist=['curvature'],
                          regmethod='chi2', chi2=f['chi_sq'], hull_vert=f['hull_vert'], rawfilename='synthetic.h5',
                          config_name='config.ini', config_path='/p', config_contents=str(f['cfg']))
    code = ("import h5py,sys\n"
            "f=h5py.File(sys.argv[1],'r')\n"
            "print(f['Coeffs/C'].shape, f['Coeffs/dC'].shape, f['UnixTime'].shape, f['FitParams/hull_vert'].shape)\n"
            "print(f['ConfigFile/Contents'].dtype.kind, f['FitParams/reglist'].dtype.kind, f['FitParams/reglist'].shape)\n"
            "print(f['Coeffs'].attrs['CLASS'], f['Coeffs'].attrs['TITLE'], f['Coeffs/C'].attrs['CLASS'], f.attrs['PYTABLES_FORMAT_VERSION'])\n"
            "print(float(f['Coeffs/C'][1,2]))\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith('PYTHON')}
    out = subprocess.run(['/opt/conda/bin/python3.9', '-c', code, fn], capture_output=True, text=True, env=env)
    if out.returncode != 0 and 'No module named' in out.stderr:
        pytest.skip('h5py not importable in the conda python')
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    T, N = f['Coeffs'].shape
    assert lines[0] == '(%d, %d) (%d, %d, %d) (%d, 2) (%d, 3)' % (T, N, T, N, N, T, f['hull_vert'].shape[0])
    assert lines[1] == 'S S (1,)'
    assert 'GROUP' in lines[2] and 'Dataset' in lines[2] and 'ARRAY' in lines[2] and '2.1' in lines[2]
    assert float(lines[3]) == f['Coeffs'][1, 2]


def _write_amisr(fn, nrec=3, nbeam=4, nrange=5, seed=0):
    rng = np.random.default_rng(seed)
    shp = (nrec, nbeam, nrange)
    ne = rng.uniform(1e10, 5e11, shp)
    dne = rng.uniform(2e10, 1e11, shp)
    chi2 = rng.uniform(0.5, 5., shp)
    fitcode = rng.integers(1, 5, shp).astype(np.int64)
    alt = rng.uniform(100e3, 600e3, (nbeam, nrange))
    alt[0, 0] = np.nan                                     # coordinate NaN -> point removed everywhere
    dne[1, 2, 3] = 5e13                                    # error above ERRLIM
    chi2[2, 1, 1] = 50.                                    # chi2 above CHI2LIM
    fitcode[0, 3, 4] = 7                                   # bad fit code
    with h5io.H5File(fn, 'w') as h5:
        for g in ('/Time', '/Geomag', '/FittedParams', '/FittedParams/FitInfo'):
            h5.create_group(g)
        t0 = 1480286700. + 60. * np.arange(nrec)
        h5.create_array('/Time/UnixTime', np.stack([t0, t0 + 60.], axis=1))
        h5.create_array('/Geomag/Altitude', alt)
        h5.create_array('/Geomag/Latitude', rng.uniform(75, 80, (nbeam, nrange)))
        h5.create_array('/Geomag/Longitude', rng.uniform(255, 270, (nbeam, nrange)))
        h5.create_array('/FittedParams/FitInfo/chi2', chi2)
        h5.create_array('/FittedParams/FitInfo/fitcode', fitcode)
        h5.create_array('/FittedParams/IonMass', np.array([16., 32.]))
        h5.create_array('/FittedParams/Ne', ne)
        h5.create_array('/FittedParams/dNe', dne)
        fits = rng.uniform(0, 1, shp + (2, 4))
        h5.create_array('/FittedParams/Fits', fits)
        h5.create_array('/FittedParams/Errors', fits * 0.1 + 2e10)
    return ne, dne, fits


def test_read_amisr_file_masks(tmp_path):
    """Quality masks of interpolate.py:645-664."""
    fn = str(tmp_path / 'amisr.h5')
    ne, dne, fits = _write_amisr(fn)
    utime, lat, lon, alt, value, error = h5io.read_amisr_file(fn, 'dens', [1e10, 1e13], [0.1, 10.], [1, 2, 3, 4])
    assert utime.shape == (3, 2) and lat.shape == lon.shape == alt.shape == (19,)       # one NaN coordinate dropped
    assert value.shape == error.shape == (3, 19)
    flat = lambda r, bm, rg: bm * 5 + rg - 1                                             # noqa: E731 (index after the drop)
    assert np.isnan(value[1, flat(1, 2, 3)]) and np.isnan(error[1, flat(1, 2, 3)])
    assert np.isnan(value[2, flat(2, 1, 1)])
    assert np.isnan(value[0, flat(0, 3, 4)])
    assert np.isfinite(value).sum() == 3 * 19 - 3
    assert value[0, 0] == ne[0, 0, 1]
    # temperature of O+ : Fits[..., m, i] with m = index of mass 16, i = 1
    _, _, _, _, v2, _ = h5io.read_amisr_file(fn, 'temp_O', [1e10, 1e13], [0.1, 10.], [1, 2, 3, 4])
    assert v2[0, 0] == fits[0, 0, 1, 0, 1]
