"""HDF5 I/O for the coefficient file and the AMISR input file.

The reference reads and writes HDF5 through PyTables (``tables``), which this image does not ship
(nor ``h5py``).  This module talks to libhdf5 directly through ctypes - a minimal subset: fixed-shape
numeric arrays, fixed-length byte strings, groups and string attributes - and lays the coefficient file
out exactly as ``Interpolate.saveh5`` does (volumetricinterp/interpolate.py:671-708):

    /UnixTime (T,2)   /Coeffs/C (T,N)   /Coeffs/dC (T,N,N)
    /FitParams/{reglist, regmethod, chi2 (T,), hull_vert (V,3)}
    /RawData/filename   /ConfigFile/{Name, Path, Contents}

with the node attributes PyTables itself writes (CLASS / VERSION / TITLE / FLAVOR), and byte strings stored as
FIXED-length HDF5 strings (PyTables cannot read variable-length ones), so that the reference's
``Estimate.loadh5`` (estimate.py:62-70) opens the files written here and vice versa.
Host-side file plumbing only; nothing here is on the GPU path.
"""
import ctypes as C
import os

import numpy as np

_CANDIDATES = [os.environ.get('VINTERP_LIBHDF5', ''), 'libhdf5.so', 'libhdf5_serial.so', '/opt/conda/lib/libhdf5.so',
               'libhdf5.so.103', 'libhdf5.so.200', 'libhdf5.so.310']
_lib = None

hid_t = C.c_int64
hsize_t = C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT = 0
H5S_ALL = 0
H5S_SCALAR = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_SGN_NONE = 0


class H5Error(IOError):
    pass


def _load():
    global _lib
    if _lib is not None:
        return _lib
    last = None
    for name in _CANDIDATES:
        if not name:
            continue
        try:
            lib = C.CDLL(name)
            break
        except OSError as e:
            last = e
    else:
        raise H5Error('libhdf5 not found (set VINTERP_LIBHDF5 to its path): %s' % last)
    lib.H5open()
    maj, mi, rel = C.c_uint(), C.c_uint(), C.c_uint()
    lib.H5get_libversion(C.byref(maj), C.byref(mi), C.byref(rel))
    if (maj.value, mi.value) < (1, 10):
        raise H5Error('libhdf5 >= 1.10 required (64-bit hid_t), found %d.%d.%d' % (maj.value, mi.value, rel.value))

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype, f.argtypes = res, list(args)
    sig('H5Fcreate', hid_t, C.c_char_p, C.c_uint, hid_t, hid_t)
    sig('H5Fopen', hid_t, C.c_char_p, C.c_uint, hid_t)
    sig('H5Fclose', C.c_int, hid_t)
    sig('H5Gcreate2', hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t)
    sig('H5Gclose', C.c_int, hid_t)
    sig('H5Oopen', hid_t, hid_t, C.c_char_p, hid_t)
    sig('H5Oclose', C.c_int, hid_t)
    sig('H5Lexists', C.c_int, hid_t, C.c_char_p, hid_t)
    sig('H5Screate_simple', hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t))
    sig('H5Screate', hid_t, C.c_int)
    sig('H5Sclose', C.c_int, hid_t)
    sig('H5Sget_simple_extent_ndims', C.c_int, hid_t)
    sig('H5Sget_simple_extent_dims', C.c_int, hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t))
    sig('H5Dcreate2', hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t)
    sig('H5Dopen2', hid_t, hid_t, C.c_char_p, hid_t)
    sig('H5Dclose', C.c_int, hid_t)
    sig('H5Dwrite', C.c_int, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
    sig('H5Dread', C.c_int, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
    sig('H5Dget_space', hid_t, hid_t)
    sig('H5Dget_type', hid_t, hid_t)
    sig('H5Tcopy', hid_t, hid_t)
    sig('H5Tclose', C.c_int, hid_t)
    sig('H5Tset_size', C.c_int, hid_t, C.c_size_t)
    sig('H5Tget_size', C.c_size_t, hid_t)
    sig('H5Tget_class', C.c_int, hid_t)
    sig('H5Tget_sign', C.c_int, hid_t)
    sig('H5Tis_variable_str', C.c_int, hid_t)
    sig('H5Acreate2', hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t)
    sig('H5Awrite', C.c_int, hid_t, hid_t, C.c_void_p)
    sig('H5Aclose', C.c_int, hid_t)
    sig('H5Eset_auto2', C.c_int, hid_t, C.c_void_p, C.c_void_p)
    lib.H5Eset_auto2(0, None, None)          # errors are reported through return codes
    _lib = lib
    return lib


def _tid(name):
    return hid_t.in_dll(_load(), name).value


_NP2H5 = {'float64': 'H5T_NATIVE_DOUBLE_g', 'float32': 'H5T_NATIVE_FLOAT_g', 'int64': 'H5T_NATIVE_LLONG_g',
          'int32': 'H5T_NATIVE_INT_g', 'int16': 'H5T_NATIVE_SHORT_g', 'int8': 'H5T_NATIVE_SCHAR_g',
          'uint64': 'H5T_NATIVE_ULLONG_g', 'uint32': 'H5T_NATIVE_UINT_g', 'uint16': 'H5T_NATIVE_USHORT_g',
          'uint8': 'H5T_NATIVE_UCHAR_g'}


def _chk(v, what):
    if v < 0:
        raise H5Error('HDF5 call failed: %s' % what)
    return v


class H5File(object):
    """Tiny read/write wrapper (context manager)."""

    def __init__(self, filename, mode='r'):
        lib = _load()
        self.lib = lib
        fn = os.fsencode(filename)
        if mode == 'w':
            self.fid = _chk(lib.H5Fcreate(fn, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), 'H5Fcreate(%s)' % filename)
            self._str_attr(self.fid, b'.', {'CLASS': 'GROUP', 'PYTABLES_FORMAT_VERSION': '2.1', 'TITLE': '',
                                            'VERSION': '1.0'})
        else:
            if not os.path.exists(filename):
                raise H5Error('no such file: %s' % filename)
            self.fid = _chk(lib.H5Fopen(fn, H5F_ACC_RDONLY, H5P_DEFAULT), 'H5Fopen(%s)' % filename)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def close(self):
        if self.fid:
            self.lib.H5Fclose(self.fid)
            self.fid = 0

    # ---- write -----------------------------------------------------------------------------------
    def _str_attr(self, loc, objname, attrs):
        lib = self.lib
        obj = _chk(lib.H5Oopen(loc, objname, H5P_DEFAULT), 'H5Oopen')
        try:
            for k, v in attrs.items():
                data = v.encode('utf-8')
                t = lib.H5Tcopy(_tid('H5T_C_S1_g'))
                lib.H5Tset_size(t, max(1, len(data)))
                sp = lib.H5Screate(H5S_SCALAR)
                a = _chk(lib.H5Acreate2(obj, k.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT), 'H5Acreate2')
                buf = C.create_string_buffer(data, max(1, len(data)))
                lib.H5Awrite(a, t, buf)
                lib.H5Aclose(a)
                lib.H5Sclose(sp)
                lib.H5Tclose(t)
        finally:
            lib.H5Oclose(obj)

    def create_group(self, path, title=''):
        g = _chk(self.lib.H5Gcreate2(self.fid, path.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 'H5Gcreate2')
        self.lib.H5Gclose(g)
        self._str_attr(self.fid, path.encode(), {'CLASS': 'GROUP', 'TITLE': title, 'VERSION': '1.0'})

    def create_array(self, path, obj):
        """PyTables ``create_array`` semantics: numeric ndarray, bytes scalar, or list of str."""
        lib = self.lib
        if isinstance(obj, (bytes, str)):
            data = obj.encode('utf-8') if isinstance(obj, str) else obj
            arr = np.array(data, dtype='S%d' % max(1, len(data)))
        elif isinstance(obj, (list, tuple)) and (len(obj) == 0 or isinstance(obj[0], (str, bytes))):
            enc = [o.encode('utf-8') if isinstance(o, str) else o for o in obj]
            arr = np.array(enc, dtype='S%d' % max([1] + [len(e) for e in enc]))
        else:
            arr = np.ascontiguousarray(obj)
        if arr.dtype.kind == 'S':
            t = lib.H5Tcopy(_tid('H5T_C_S1_g'))
            lib.H5Tset_size(t, arr.dtype.itemsize)
            own_t = True
        else:
            key = arr.dtype.name
            if key not in _NP2H5:
                raise H5Error('unsupported dtype %s for %s' % (arr.dtype, path))
            t = _tid(_NP2H5[key])
            own_t = False
        if arr.ndim == 0:
            sp = lib.H5Screate(H5S_SCALAR)
        else:
            dims = (hsize_t * arr.ndim)(*arr.shape)
            sp = lib.H5Screate_simple(arr.ndim, dims, None)
        d = _chk(lib.H5Dcreate2(self.fid, path.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT),
                 'H5Dcreate2(%s)' % path)
        arr = np.ascontiguousarray(arr)
        if arr.size:
            _chk(lib.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)), 'H5Dwrite(%s)' % path)
        lib.H5Dclose(d)
        lib.H5Sclose(sp)
        if own_t:
            lib.H5Tclose(t)
        self._str_attr(self.fid, path.encode(), {'CLASS': 'ARRAY', 'FLAVOR': 'numpy', 'TITLE': '', 'VERSION': '2.4'})

    # ---- read ------------------------------------------------------------------------------------
    def exists(self, path):
        parts = [p for p in path.split('/') if p]
        cur = ''
        for p in parts:
            cur += '/' + p
            if self.lib.H5Lexists(self.fid, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def read(self, path):
        """Whole dataset as an ndarray (numeric) or bytes / array of bytes (fixed-length strings)."""
        lib = self.lib
        if not self.exists(path):
            raise H5Error('no such node: %s' % path)
        d = _chk(lib.H5Dopen2(self.fid, path.encode(), H5P_DEFAULT), 'H5Dopen2(%s)' % path)
        try:
            sp = lib.H5Dget_space(d)
            nd = lib.H5Sget_simple_extent_ndims(sp)
            dims = (hsize_t * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(sp, dims, None)
            shape = tuple(int(dims[i]) for i in range(nd))
            lib.H5Sclose(sp)
            t = lib.H5Dget_type(d)
            cls, size = lib.H5Tget_class(t), lib.H5Tget_size(t)
            if cls == H5T_STRING:
                if lib.H5Tis_variable_str(t) > 0:
                    lib.H5Tclose(t)
                    raise H5Error('%s: variable-length strings are not supported' % path)
                out = np.empty(shape, dtype='S%d' % size)
                mt = lib.H5Tcopy(_tid('H5T_C_S1_g'))
                lib.H5Tset_size(mt, size)
                if out.size:
                    _chk(lib.H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)), 'H5Dread')
                lib.H5Tclose(mt)
                lib.H5Tclose(t)
                return out[()] if nd == 0 else out
            if cls == H5T_FLOAT:
                dt = {8: 'float64', 4: 'float32'}[size]
            elif cls == H5T_INTEGER:
                unsigned = lib.H5Tget_sign(t) == H5T_SGN_NONE
                dt = ('uint%d' if unsigned else 'int%d') % (8 * size)
            else:
                lib.H5Tclose(t)
                raise H5Error('%s: unsupported HDF5 type class %d' % (path, cls))
            lib.H5Tclose(t)
            out = np.empty(shape, dtype=dt)
            if out.size:
                _chk(lib.H5Dread(d, _tid(_NP2H5[dt]), H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)),
                     'H5Dread(%s)' % path)
            return out
        finally:
            lib.H5Dclose(d)


# ----------------------------------------------------------------------------------------------------
def write_coeff_file(filename, time, Coeffs, Covariance, reglist, regmethod, chi2, hull_vert, rawfilename,
                     config_name, config_path, config_contents):
    """Layout of Interpolate.saveh5, interpolate.py:680-708."""
    with H5File(filename, 'w') as h5:
        h5.create_group('/Coeffs', 'Dataset')
        h5.create_group('/FitParams', 'Dataset')
        h5.create_group('/RawData', 'Dataset')
        h5.create_array('/UnixTime', np.asarray(time))
        h5.create_array('/Coeffs/C', np.asarray(Coeffs, dtype=np.float64))
        h5.create_array('/Coeffs/dC', np.asarray(Covariance, dtype=np.float64))
        h5.create_array('/FitParams/reglist', list(reglist))
        h5.create_array('/FitParams/regmethod', regmethod.encode('utf-8'))
        h5.create_array('/FitParams/chi2', np.asarray(chi2, dtype=np.float64))
        h5.create_array('/FitParams/hull_vert', np.asarray(hull_vert, dtype=np.float64))
        h5.create_array('/RawData/filename', rawfilename.encode('utf-8'))
        h5.create_group('/ConfigFile', '')
        h5.create_array('/ConfigFile/Name', config_name.encode('utf-8'))
        h5.create_array('/ConfigFile/Path', config_path.encode('utf-8'))
        h5.create_array('/ConfigFile/Contents', config_contents.encode('utf-8'))


def read_coeff_file(filename):
    """What Estimate.loadh5 reads, estimate.py:62-70."""
    with H5File(filename, 'r') as h5:
        txt = h5.read('/ConfigFile/Contents')
        return dict(Coeffs=h5.read('/Coeffs/C'), Covariance=h5.read('/Coeffs/dC'), time=h5.read('/UnixTime'),
                    hull_vert=h5.read('/FitParams/hull_vert'),
                    config_file_text=bytes(txt) if not isinstance(txt, bytes) else txt)


INDEX_DICT = {'frac': 0, 'temp': 1, 'colfreq': 2}                 # interpolate.py:605
MASS_DICT = {'O': 16, 'O2': 32, 'NO': 30, 'N2': 28, 'N': 14}      # interpolate.py:606


def read_amisr_file(filename, param, errlim, chi2lim, goodfitcode):
    """AMISR fitted-file reader + quality masks, interpolate.py:582-667."""
    with H5File(filename, 'r') as h5:
        utime = h5.read('/Time/UnixTime')
        alt = h5.read('/Geomag/Altitude')
        lat = h5.read('/Geomag/Latitude')
        lon = h5.read('/Geomag/Longitude')
        c2 = h5.read('/FittedParams/FitInfo/chi2')
        fc = h5.read('/FittedParams/FitInfo/fitcode')
        imass = h5.read('/FittedParams/IonMass')
        if param == 'dens':
            val = h5.read('/FittedParams/Ne')
            err = h5.read('/FittedParams/dNe')
        else:
            parts = param.split('_')
            i = INDEX_DICT[parts[0]]
            hit = np.where(imass == MASS_DICT[parts[1]])[0]
            m = int(hit[0]) if hit.size else -1
            val = h5.read('/FittedParams/Fits')[:, :, :, m, i]
            err = h5.read('/FittedParams/Errors')[:, :, :, m, i]
    altitude, latitude, longitude = alt.flatten(), lat.flatten(), lon.flatten()
    chi2 = c2.reshape(c2.shape[0], -1)
    fitcode = fc.reshape(fc.shape[0], -1)
    value = np.array(val.reshape(val.shape[0], -1), dtype=np.float64)
    error = np.array(err.reshape(err.shape[0], -1), dtype=np.float64)
    if np.nanmedian(chi2) > 100.:            # some files over-estimate chi2 by 369 (interpolate.py:645-646)
        chi2 = chi2 - 369.
    with np.errstate(invalid='ignore'):
        checks = np.array([error > errlim[0], error < errlim[1], chi2 > chi2lim[0], chi2 < chi2lim[1],
                           np.isin(fitcode, goodfitcode)])
    bad = np.squeeze(np.any(checks == False, axis=0, keepdims=True))      # noqa: E712
    value[bad] = np.nan
    error[bad] = np.nan
    fin = np.isfinite(altitude)
    return (utime, latitude[fin], longitude[fin], altitude[fin], value[:, fin], error[:, fin])
