"""Laguerre x spherical-cap-harmonic model on the MI355X.

Host-side mirror of the reference plug-in ``volumetricinterp/models/sphharmlag.py``
(class ``Model``, :18-359): same constructor (a config *file object*), same
attributes (``maxk, maxl, nbasis, cap_lim, latcp, loncp, max_z_int``), same
methods (``basis_numbers, nu, Kvm, Az, transform_coord, basis``) and the same
``eval_reg_matricies`` dictionary.  ``basis`` / ``transform_coord`` run on the
GPU through libvinterp.so (kernels in csrc/vi_basis.hip); what stays on the
host is the one-off table construction below and the regularisation matrices
(quadrature of QUADPACK artefacts, not a GPU target).

How P_nu^m(cos theta) (``scipy.special.lpmv`` at sphharmlag.py:141) is mapped
to the device: degrees nu_l = nv_l + v0 sharing a fractional part v0 form a
group; per group and order m one upward degree recurrence (AMS 8.5.3, the one
lpmv itself uses) runs from seeds at degrees v0+m, v0+m+1 - closed form for
integer degree, the 2F1((1-x)/2) series otherwise - in the normalised form
``p_j = x p_{j-1} - c_j p_{j-2}`` (two fp64 ops per step); the normalisation,
``Kvm`` and lpmv's negative-order Gamma ratio are folded into one
constant per (l, signed m).
"""
import configparser
import ctypes as C
import math

import numpy as np
import scipy.special as sp

from .. import _lib
from ..geodesy import geodetic2ecef
from ._base import DeviceModel

RE = 6371.2 * 1000.           # Earth radius (m), sphharmlag.py:9

SNAP_TOL = 1e-12              # degrees whose fractional parts differ by less share one recurrence
HYP_TERMS = 256               # 2F1 series table length (converges for colatitudes < ~135 deg)


class Model(DeviceModel):
    def __init__(self, config_file, ctx=None):
        self.read_config(config_file)
        self.nbasis = self.maxk * self.maxl**2
        self.cap_lim = self.cap_lim * np.pi / 180.
        self.eval_reg_matricies = {'curvature': self.eval_omega, '0thorder': self.eval_psi}
        self._ctx = ctx
        self._handle = None
        self._keep = None

    # sphharmlag.py:65-75
    def read_config(self, config_file):
        config = configparser.ConfigParser()
        config.read_file(config_file)
        self.maxk = config.getint('MODEL', 'MAXK')
        self.maxl = config.getint('MODEL', 'MAXL')
        self.latcp = config.getfloat('MODEL', 'LATCP')
        self.loncp = config.getfloat('MODEL', 'LONCP')
        self.cap_lim = config.getfloat('MODEL', 'CAP_LIM')
        self.max_z_int = float(config.get('MODEL', 'MAX_Z_INT'))

    # sphharmlag.py:79-99 (l, m come back as float64, as in the reference)
    def basis_numbers(self, n):
        k = n // (self.maxl**2)
        r = n % (self.maxl**2)
        l = np.floor(np.sqrt(r))
        m = r - l * (l + 1)
        return k, l, m

    # sphharmlag.py:101-115
    def nu(self, n):
        k, l, m = self.basis_numbers(n)
        return (2 * l + 0.5) * np.pi / (2 * self.cap_lim) - 0.5

    # sphharmlag.py:305-321 (Gamma overflow -> 0 / NaN kept)
    def Kvm(self, v, m):
        with np.errstate(all='ignore'):
            K = np.sqrt((2 * v + 1) / (4 * np.pi) * sp.gamma(float(v - m + 1)) / sp.gamma(float(v + m + 1)))
        if m != 0:
            K = K * np.sqrt(2)
        return K

    # sphharmlag.py:263-281
    def Az(self, v, m, phi):
        if m < 0:
            return self.Kvm(v, abs(m)) * np.sin(abs(m) * phi)
        return self.Kvm(v, abs(m)) * np.cos(abs(m) * phi)

    # ------------------------------------------------------------------------------------------
    def _rotation(self):
        """theta0, phi0 and axis k of sphharmlag.py:345-349."""
        x0, y0, z0 = geodetic2ecef(self.latcp, self.loncp, 0.)
        theta0 = np.arccos(z0 / np.sqrt(x0**2 + y0**2 + z0**2))
        phi0 = np.arctan2(y0, x0)
        return float(theta0), float(np.cos(phi0 + np.pi / 2.)), float(np.sin(phi0 + np.pi / 2.))

    def device_tables(self):
        """Build the vi_model_desc tables (pure host arithmetic; unit-tested without a GPU)."""
        maxl, maxk = self.maxl, self.maxk
        nus = np.array([(2 * np.float64(l) + 0.5) * np.pi / (2 * self.cap_lim) - 0.5 for l in range(maxl)])
        if np.any(nus < 0):
            raise ValueError('CAP_LIM > 90 deg gives negative degrees; not supported')
        nvs = np.floor(nus).astype(np.int64)
        v0s = nus - nvs
        # group degrees by fractional part
        groups = []          # list of dict(v0=, members=[l...])
        for l in range(maxl):
            for g in groups:
                if abs(g['v0'] - v0s[l]) < SNAP_TOL:
                    g['members'].append(l)
                    break
            else:
                groups.append(dict(v0=float(v0s[l]), members=[l]))
        scale = np.zeros(maxl * maxl)
        scale1 = np.zeros(maxl * maxl)        # same for degree nu_l + 1 (grad_basis needs lpmv(m, nu+1, x))
        out_groups = []
        for g in groups:
            v0 = np.longdouble(g['v0'])
            nvmax = int(max(nvs[l] for l in g['members']))
            pick = np.full(nvmax + 2, -1, dtype=np.int32)
            for l in g['members']:
                pick[nvs[l]] = l
            # one row more than the largest degree: grad_basis also needs the degrees nu_l + 1
            c = np.zeros((nvmax + 2, maxl))
            s = np.ones((nvmax + 2, maxl), dtype=np.longdouble)      # chain normalisation s[j][m]
            for m in range(maxl):
                for j in range(m + 2, nvmax + 2):
                    a = (2 * (v0 + j) - 1) / (v0 + j - m)
                    b = (v0 + j - 1 + m) / (v0 + j - m)
                    s[j, m] = s[j - 1, m] * a
                    c[j, m] = float(b * s[j - 2, m] / s[j, m])
            for l in g['members']:
                v = float(nus[l])
                for m in range(0, l + 1):
                    K = self.Kvm(v, m)
                    sjm = float(s[nvs[l], m])
                    sjm1 = float(s[nvs[l] + 1, m])
                    scale[l * (l + 1) + m] = K * sjm
                    scale1[l * (l + 1) + m] = K * sjm1
                    if m > 0:
                        # scipy.special.lpmv for negative order: (-1)^m Gamma(v-m+1)/Gamma(v+m+1) P_v^m
                        with np.errstate(all='ignore'):
                            neg = (-1.)**m * sp.gamma(v - m + 1) / sp.gamma(v + m + 1)
                            neg1 = (-1.)**m * sp.gamma(v + 1 - m + 1) / sp.gamma(v + 1 + m + 1)
                            scale[l * (l + 1) - m] = K * neg * sjm
                            scale1[l * (l + 1) - m] = K * neg1 * sjm1
            entry = dict(v0=float(g['v0']), nvmax=nvmax, pick=pick, c=np.ascontiguousarray(c), nterms=0,
                         pref=None, q=None, members=list(g['members']))
            if g['v0'] != 0.0:
                nt = HYP_TERMS
                pref = np.zeros((2, maxl))
                q = np.zeros((2, maxl, nt))
                i = np.arange(nt, dtype=np.float64)
                fv0 = float(g['v0'])
                for m in range(maxl):
                    for which in (0, 1):
                        v = fv0 + m + which
                        a_, b_, c_ = m - v, m + v + 1., m + 1.
                        with np.errstate(all='ignore'):
                            pref[which, m] = ((-1.)**m * sp.gamma(v + m + 1) /
                                              (sp.gamma(v - m + 1) * math.factorial(m) * 2.**m))
                        q[which, m] = (a_ + i) * (b_ + i) / ((c_ + i) * (i + 1.))
                entry.update(nterms=nt, pref=np.ascontiguousarray(pref), q=np.ascontiguousarray(q))
            out_groups.append(entry)
        theta0, kx, ky = self._rotation()
        return dict(groups=out_groups, scale=scale, scale1=scale1, rot_cos=math.cos(theta0), rot_sin=math.sin(theta0),
                    kx=kx, ky=ky, nus=nus, nvs=nvs)

    def _create_handle(self, ctx):
        tb = self.device_tables()
        garr = (_lib.SphGroup * len(tb['groups']))()
        for i, g in enumerate(tb['groups']):
            garr[i].v0 = g['v0']
            garr[i].nvmax = g['nvmax']
            garr[i].nterms = g['nterms']
            garr[i].pick = g['pick'].ctypes.data_as(_lib.c_int32_p)
            garr[i].c = g['c'].ctypes.data_as(_lib.c_double_p)
            if g['nterms']:
                garr[i].seed_pref = g['pref'].ctypes.data_as(_lib.c_double_p)
                garr[i].seed_q = g['q'].ctypes.data_as(_lib.c_double_p)
        d = _lib.ModelDesc()
        d.kind = _lib.VI_MODEL_SPHHARMLAG
        d.nbasis = self.nbasis
        d.maxk, d.maxl = self.maxk, self.maxl
        d.rot_cos, d.rot_sin = tb['rot_cos'], tb['rot_sin']
        d.rot_kx, d.rot_ky = tb['kx'], tb['ky']
        d.earth_radius = RE
        d.ngroups = len(tb['groups'])
        d.groups = garr
        d.coef_scale = tb['scale'].ctypes.data_as(_lib.c_double_p)
        d.coef_scale1 = tb['scale1'].ctypes.data_as(_lib.c_double_p)
        tb['nus'] = np.ascontiguousarray(tb['nus'], dtype=np.float64)
        d.nu = tb['nus'].ctypes.data_as(_lib.c_double_p)
        h = _lib.VOIDP()
        _lib.check(_lib.lib.vi_model_create(ctx.handle, C.byref(d), C.byref(h)), 'vi_model_create')
        return h, (tb, garr)

    # sphharmlag.py:324-359
    def transform_coord(self, gdlat, gdlon, gdalt):
        shape = np.asarray(gdlat).shape
        return tuple(o.reshape(shape) for o in self._transform(gdlat, gdlon, gdalt))

    # sphharmlag.py:284-302
    def dAz(self, v, m, phi):
        if m < 0:
            return abs(m) * self.Kvm(v, abs(m)) * np.cos(abs(m) * phi)
        return -1 * m * self.Kvm(v, abs(m)) * np.sin(abs(m) * phi)

    # sphharmlag.py:148-184 (advertised by the reference, never called by its own workflow - SURVEY F9, row N1)
    def grad_basis(self, gdlat, gdlon, gdalt):
        """Gradient of every basis function: (P, 3, N), components along z, theta, phi; 1-D inputs as in the
        reference."""
        gdlat = np.asarray(gdlat, dtype=np.float64)
        P, N = gdlat.size, self.nbasis
        if P == 0:
            return np.zeros((0, 3, N))
        h = self.handle()
        dlat, dlon, dalt = self._upload_coords(gdlat, gdlon, gdalt)
        dG = self._ctx.empty((P, 3, N))
        _lib.check(_lib.lib.vi_grad_basis_f64(h, P, dlat.ptr, dlon.ptr, dalt.ptr, dG.ptr, 3 * N, N, 1),
                   'vi_grad_basis_f64')
        return dG.download()

    # ---- regularisation matrices (host) -------------------------------------------
    def eval_omega(self):
        from ..regmat import eval_omega
        return eval_omega(self)

    def eval_psi(self):
        # The reference's own values by default (regmat.eval_psi: its integrands and quad calls, each distinct 1-D integral
        # once - bit-identical to sphharmlag.py:215-239, 2.4 s at N = 144).  VINTERP_REGMAT=gauss: exact quadrature
        # (regmat.eval_psi_gauss: milliseconds; 5e-10 of max|Psi| away from the QUADPACK values, which moves the fitted
        # coefficients of tests/golden/fit_k8l2_psi.npz by less than 1e-6 - tests/test_gpu_fit.py - but a default must give
        # the reference's numbers).
        import os
        from ..regmat import eval_psi, eval_psi_gauss
        if os.environ.get('VINTERP_REGMAT', 'quad') == 'gauss':
            return eval_psi_gauss(self)
        return eval_psi(self)
