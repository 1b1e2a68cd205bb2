"""Oracle (test infrastructure): Gaussian radial-basis-function model.

Restates ``volumetricinterp/models/radbasfun.py`` of the reference
(``Model.__init__`` :51-62, ``read_config`` :65-78, ``basis`` :83-112,
``transform_coords`` :232-256).  Centres come from ``np.meshgrid`` with its
default 'xy' indexing, exactly as the reference builds them.
"""
import configparser

import numpy as np

from .geodesy import geodetic2ecef


class RadBasFunOracle:
    def __init__(self, eps=100000.0, latrange=(74., 80.), lonrange=(260., 285.),
                 altrange=(100., 600.), numgridpnt=7, latcp=78.0, loncp=262.0):
        self.eps = float(eps)
        self.latrange = [float(v) for v in latrange]
        self.lonrange = [float(v) for v in lonrange]
        self.altrange = [float(v) for v in altrange]
        self.numgridpnt = int(numgridpnt)
        self.latcp, self.loncp = float(latcp), float(loncp)
        # radbasfun.py:55-60
        g = self.numgridpnt
        lat, lon, alt = np.meshgrid(np.linspace(self.latrange[0], self.latrange[1], g),
                                    np.linspace(self.lonrange[0], self.lonrange[1], g),
                                    np.linspace(self.altrange[0], self.altrange[1], g) * 1000.)
        X, Y, Z = geodetic2ecef(lat.flatten(), lon.flatten(), alt.flatten())
        self.centers = np.array([X, Y, Z]).T
        self.nbasis = self.centers.shape[0]

    @classmethod
    def from_config(cls, fileobj):
        cfg = configparser.ConfigParser()
        cfg.read_file(fileobj)
        g = lambda k: cfg.get('MODEL', k)
        rng = lambda k: [float(i) for i in g(k).split(',')]
        return cls(eps=float(g('EPS')), latrange=rng('LATRANGE'), lonrange=rng('LONRANGE'),
                   altrange=rng('ALTRANGE'), numgridpnt=int(g('NUMGRIDPNT')),
                   latcp=float(g('LATCP')), loncp=float(g('LONCP')))

    # radbasfun.py:83-112
    def basis(self, gdlat, gdlon, gdalt):
        gdlat = np.asarray(gdlat, dtype=np.float64)
        gdlon = np.asarray(gdlon, dtype=np.float64)
        gdalt = np.asarray(gdalt, dtype=np.float64)
        x, y, z = geodetic2ecef(gdlat.flatten(), gdlon.flatten(), gdalt.flatten())
        R = np.array([x, y, z])                                 # (3, P)
        cols = []
        for n in range(self.nbasis):
            c = self.centers[n]
            r = np.linalg.norm(R - c[:, None], axis=0)
            cols.append(np.exp(-r**2 / self.eps**2))
        A = np.array(cols)
        return np.moveaxis(A.reshape((-1,) + gdlat.shape), 0, -1)
