"""Oracle (test infrastructure): Laguerre x spherical-cap-harmonic basis.

NumPy/SciPy restatement of ``volumetricinterp/models/sphharmlag.py`` of the
reference (class ``Model``, lines 18-359).  Same algorithmic structure as the
reference: one special-function evaluation per basis index n, no reuse across
indices sharing k or (l, m).  Known quirks reproduced on purpose
(SURVEY.md F3, F4, F8):
  * the Rodrigues rotation uses +theta0 (sphharmlag.py:353),
  * the *signed* m goes into ``scipy.special.lpmv`` while ``Kvm`` takes |m|
    (sphharmlag.py:141, :278-281),
  * ``Kvm`` is formed from a ratio of ``scipy.special.gamma`` values and
    overflows to 0 / NaN for large degree (sphharmlag.py:318).
"""
import configparser

import numpy as np
import scipy.special as sp

from .geodesy import geodetic2ecef

RE = 6371.2 * 1000.0          # sphharmlag.py:9


class SphHarmLagOracle:
    def __init__(self, maxk=4, maxl=6, cap_lim_deg=10.0, latcp=78.0, loncp=262.0,
                 max_z_int=float('inf')):
        # sphharmlag.py:57-62, :70-75
        self.maxk = int(maxk)
        self.maxl = int(maxl)
        self.latcp = float(latcp)
        self.loncp = float(loncp)
        self.max_z_int = float(max_z_int)
        self.nbasis = self.maxk * self.maxl**2
        self.cap_lim = float(cap_lim_deg) * np.pi / 180.0

    @classmethod
    def from_config(cls, fileobj):
        cfg = configparser.ConfigParser()
        cfg.read_file(fileobj)
        g = lambda k: cfg.get('MODEL', k)
        return cls(maxk=int(g('MAXK')), maxl=int(g('MAXL')), cap_lim_deg=float(g('CAP_LIM')),
                   latcp=float(g('LATCP')), loncp=float(g('LONCP')), max_z_int=float(g('MAX_Z_INT')))

    # sphharmlag.py:79-99
    def basis_numbers(self, n):
        l2 = self.maxl**2
        k = n // l2
        r = n % l2
        l = np.floor(np.sqrt(r))
        m = r - l * (l + 1)
        return k, l, m

    # sphharmlag.py:101-115
    def nu(self, n):
        _, l, _ = self.basis_numbers(n)
        return (2 * l + 0.5) * np.pi / (2 * self.cap_lim) - 0.5

    # sphharmlag.py:305-321
    def Kvm(self, v, m):
        with np.errstate(all='ignore'):
            K = np.sqrt((2 * v + 1) / (4 * np.pi) * sp.gamma(float(v - m + 1)) / sp.gamma(float(v + m + 1)))
        if m != 0:
            K = K * np.sqrt(2)
        return K

    # sphharmlag.py:263-281
    def Az(self, v, m, phi):
        am = abs(m)
        if m < 0:
            return self.Kvm(v, am) * np.sin(am * phi)
        return self.Kvm(v, am) * np.cos(am * phi)

    # sphharmlag.py:324-359
    def transform_coord(self, gdlat, gdlon, gdalt):
        x0, y0, z0 = geodetic2ecef(self.latcp, self.loncp, 0.0)
        theta0 = np.arccos(z0 / np.sqrt(x0**2 + y0**2 + z0**2))
        phi0 = np.arctan2(y0, x0)
        k = np.array([np.cos(phi0 + np.pi / 2.0), np.sin(phi0 + np.pi / 2.0), 0.0])

        x, y, z = geodetic2ecef(gdlat, gdlon, gdalt)
        R = np.stack([np.atleast_1d(x), np.atleast_1d(y), np.atleast_1d(z)], axis=1)   # (P,3)
        # Rodrigues with +theta0 exactly as sphharmlag.py:353 (vectorised over points;
        # each output element is the same three-term sum in the same order)
        kdotR = k[0] * R[:, 0] + k[1] * R[:, 1] + k[2] * R[:, 2]
        Rr = (R * np.cos(theta0) + np.cross(k[None, :], R) * np.sin(theta0)
              + k[None, :] * kdotR[:, None] * (1 - np.cos(theta0)))
        r = np.sqrt(Rr[:, 0]**2 + Rr[:, 1]**2 + Rr[:, 2]**2)
        t = np.arccos(Rr[:, 2] / r)
        p = np.arctan2(Rr[:, 1], Rr[:, 0])
        return 100 * (r / RE - 1), t, p

    # sphharmlag.py:118-145
    def basis(self, gdlat, gdlon, gdalt):
        gdlat = np.asarray(gdlat, dtype=np.float64)
        gdlon = np.asarray(gdlon, dtype=np.float64)
        gdalt = np.asarray(gdalt, dtype=np.float64)
        z, theta, phi = self.transform_coord(gdlat.flatten(), gdlon.flatten(), gdalt.flatten())
        cols = []
        ct = np.cos(theta)
        with np.errstate(all='ignore'):
            for n in range(self.nbasis):
                k, l, m = self.basis_numbers(n)
                v = self.nu(n)
                cols.append(np.exp(-0.5 * z) * sp.eval_laguerre(k, z) * self.Az(v, m, phi) * sp.lpmv(m, v, ct))
        A = np.array(cols)                                     # (N, P)
        return np.moveaxis(A.reshape((-1,) + gdlat.shape), 0, -1)

    # sphharmlag.py:284-302
    def dAz(self, v, m, phi):
        am = abs(m)
        if m < 0:
            return am * self.Kvm(v, am) * np.cos(am * phi)
        return -1 * m * self.Kvm(v, am) * np.sin(am * phi)

    # sphharmlag.py:148-184 (never called by the reference's own workflow; next row N1).
    # 1-D inputs; returns (P, 3, N): components along z, theta, phi of the gradient of every basis function.
    def grad_basis(self, gdlat, gdlon, gdalt):
        z, theta, phi = self.transform_coord(gdlat, gdlon, gdalt)
        rows = []
        x = np.cos(theta)
        y = np.sin(theta)
        e = np.exp(-0.5 * z)
        with np.errstate(all='ignore'):
            for n in range(self.nbasis):
                k, l, m = self.basis_numbers(n)
                v = self.nu(n)
                L0 = sp.eval_laguerre(k, z)
                L1 = sp.eval_genlaguerre(k - 1, 1, z)
                Pmv = sp.lpmv(m, v, x)
                Pmv1 = sp.lpmv(m, v + 1, x)
                A = self.Az(v, m, phi)
                zhat = -0.5 * e * (L0 + 2 * L1) * Pmv * A * 100. / RE
                that = e * L0 * (-(v + 1) * x * Pmv + (v - m + 1) * Pmv1) * A / (y * (z / 100. + 1) * RE)
                phat = e * L0 * Pmv * self.dAz(v, m, phi) / (y * (z / 100. + 1) * RE)
                rows.append([zhat, that, phat])
        return np.array(rows).T
