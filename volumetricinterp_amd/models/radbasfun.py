"""Gaussian radial-basis-function model on the MI355X.

Host-side mirror of the reference plug-in ``volumetricinterp/models/radbasfun.py``
(``Model.__init__`` :51-62, ``read_config`` :65-78, ``basis`` :83-112,
``transform_coords`` :232-256).  ``basis`` runs on the GPU (csrc/vi_basis.hip,
k_basis_rbf / k_eval_rbf); the centre grid is a host one-off.
"""
import configparser
import ctypes as C

import numpy as np

from .. import _lib
from ..geodesy import geodetic2ecef

RE = 6371.2 * 1000.


class Model(object):
    def __init__(self, config_file, ctx=None):
        self.read_config(config_file)
        # radbasfun.py:55-60: default 'xy' meshgrid indexing, altitudes in km -> m
        lat, lon, alt = np.meshgrid(np.linspace(self.latrange[0], self.latrange[1], self.numgridpnt),
                                    np.linspace(self.lonrange[0], self.lonrange[1], self.numgridpnt),
                                    np.linspace(self.altrange[0], self.altrange[1], self.numgridpnt) * 1000.)
        X, Y, Z = geodetic2ecef(lat.flatten(), lon.flatten(), alt.flatten())
        self.centers = np.array([X, Y, Z]).T
        self.nbasis = self.centers.shape[0]
        self.eval_reg_matricies = {}
        self._ctx = ctx
        self._handle = None
        self._keep = None

    def read_config(self, config_file):
        config = configparser.ConfigParser()
        config.read_file(config_file)
        self.latcp = config.getfloat('MODEL', 'LATCP')
        self.loncp = config.getfloat('MODEL', 'LONCP')
        self.eps = config.getfloat('MODEL', 'EPS')
        self.latrange = [float(i) for i in config.get('MODEL', 'LATRANGE').split(',')]
        self.lonrange = [float(i) for i in config.get('MODEL', 'LONRANGE').split(',')]
        self.altrange = [float(i) for i in config.get('MODEL', 'ALTRANGE').split(',')]
        self.numgridpnt = config.getint('MODEL', 'NUMGRIDPNT')

    def handle(self, ctx=None):
        if self._handle is not None:
            return self._handle
        if ctx is not None:
            self._ctx = ctx
        if self._ctx is None:
            self._ctx = _lib.get_context()
        cen = np.ascontiguousarray(self.centers, dtype=np.float64)
        d = _lib.ModelDesc()
        d.kind = _lib.VI_MODEL_RADBASFUN
        d.nbasis = self.nbasis
        d.centers = cen.ctypes.data_as(_lib.c_double_p)
        d.eps = self.eps
        h = _lib.VOIDP()
        _lib.check(_lib.lib.vi_model_create(self._ctx.handle, C.byref(d), C.byref(h)), 'vi_model_create')
        self._handle = h
        self._keep = cen
        return h

    @property
    def ctx(self):
        self.handle()
        return self._ctx

    def __del__(self):
        try:
            if self._handle is not None and self._ctx is not None and self._ctx.handle:
                _lib.lib.vi_model_destroy(self._handle)
        except Exception:
            pass
        self._handle = None

    def _upload_coords(self, gdlat, gdlon, gdalt):
        ctx = self.ctx
        return (ctx.to_device(np.asarray(gdlat, dtype=np.float64).ravel()),
                ctx.to_device(np.asarray(gdlon, dtype=np.float64).ravel()),
                ctx.to_device(np.asarray(gdalt, dtype=np.float64).ravel()))

    # radbasfun.py:232-256: ECEF coordinates, (3, P)
    def transform_coords(self, lat, lon, alt):
        h = self.handle()
        lat = np.asarray(lat, dtype=np.float64)
        P = lat.size
        dlat, dlon, dalt = self._upload_coords(lat, lon, alt)
        out = [self._ctx.empty(P) for _ in range(3)]
        _lib.check(_lib.lib.vi_transform_f64(h, P, dlat.ptr, dlon.ptr, dalt.ptr, out[0].ptr, out[1].ptr, out[2].ptr),
                   'vi_transform_f64')
        return np.array([o.download() for o in out])

    def basis_device(self, dlat, dlon, dalt, P, transposed=False):
        h = self.handle()
        N = self.nbasis
        dA = self._ctx.empty((N, P) if transposed else (P, N))
        ld_p, ld_n = (1, P) if transposed else (N, 1)
        _lib.check(_lib.lib.vi_basis_f64(h, P, dlat.ptr, dlon.ptr, dalt.ptr, dA.ptr, ld_p, ld_n), 'vi_basis_f64')
        return dA

    # radbasfun.py:83-112
    def basis(self, gdlat, gdlon, gdalt):
        gdlat = np.asarray(gdlat, dtype=np.float64)
        P = gdlat.size
        if P == 0:
            return np.zeros(gdlat.shape + (self.nbasis,))
        dlat, dlon, dalt = self._upload_coords(gdlat, gdlon, gdalt)
        A = self.basis_device(dlat, dlon, dalt, P).download()
        return A.reshape(gdlat.shape + (self.nbasis,))
