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
from ._base import DeviceModel

RE = 6371.2 * 1000.


class Model(DeviceModel):
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

    def _create_handle(self, ctx):
        cen = np.ascontiguousarray(self.centers, dtype=np.float64)
        d = _lib.ModelDesc()
        d.kind = _lib.VI_MODEL_RADBASFUN
        d.nbasis = self.nbasis
        d.centers = cen.ctypes.data_as(_lib.c_double_p)
        d.eps = self.eps
        h = _lib.VOIDP()
        _lib.check(_lib.lib.vi_model_create(ctx.handle, C.byref(d), C.byref(h)), 'vi_model_create')
        return h, cen

    # radbasfun.py:232-256: ECEF coordinates, (3, P)
    def transform_coords(self, lat, lon, alt):
        return np.array(self._transform(lat, lon, alt))
