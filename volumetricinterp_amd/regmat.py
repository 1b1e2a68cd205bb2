"""Regularisation matrices Omega (curvature) and Psi (0th order) of the sphharmlag model.

Host-side, once per run (not a GPU target).  Restates
``eval_omega/omega_ij`` and ``eval_psi/psi_ij`` of the reference
(volumetricinterp/models/sphharmlag.py:188-239): each entry is a product of three
one-dimensional ``scipy.integrate.quad`` integrals over z, theta and phi.  The integrands and quad
calls are the reference's, so every entry has the reference's value (including the QUADPACK
artefacts of the divergent z-integral of Omega); the only change is that each distinct
1-D integral is evaluated once instead of once per (ni, nj) pair: the z-integral depends only on
(ki, kj), the theta- and phi-integrals only on the ordered pair of (l, m) indices - 10 + 2 x 1 296
quadratures at the default order instead of 31 320.
"""
import warnings

import numpy as np
import scipy.integrate
import scipy.special as sp


def _assemble(model, z_int, t_int):
    N = model.nbasis
    L2 = model.maxl**2
    out = np.zeros((N, N))
    Iz_cache, Itp_cache = {}, {}
    klm = [model.basis_numbers(n) for n in range(N)]
    nus = [model.nu(n) for n in range(N)]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')          # IntegrationWarning: the reference emits (and ignores) them too
        for ni in range(N):
            ki, li, mi = klm[ni]
            vi = nus[ni]
            for nj in range(ni, N):
                kj, lj, mj = klm[nj]
                vj = nus[nj]
                kz = (ki, kj)
                if kz not in Iz_cache:
                    Iz_cache[kz] = scipy.integrate.quad(z_int(ki, kj), 0., model.max_z_int)[0]
                kt = (ni % L2, nj % L2)
                if kt not in Itp_cache:
                    It = scipy.integrate.quad(t_int(mi, vi, mj, vj), 0., model.cap_lim)[0]
                    Ip = scipy.integrate.quad(lambda p: model.Az(vi, mi, p) * model.Az(vj, mj, p), 0., 2 * np.pi)[0]
                    Itp_cache[kt] = (It, Ip)
                It, Ip = Itp_cache[kt]
                v = Iz_cache[kz] * It * Ip
                out[ni, nj] = v
                out[nj, ni] = v
    return out


# sphharmlag.py:198-212
def eval_omega(model):
    def z_int(ki, kj):
        return lambda z: np.exp(-1 * z) * sp.eval_laguerre(ki, z) * sp.eval_laguerre(kj, z) / z**2

    def lap(m, v, t):
        c = np.cos(t)
        return (-1 * v * (v * c**2 + v + 1) * sp.lpmv(m, v, c) + v * (v + m) * c * sp.lpmv(m, v - 1, c)
                + v * (v - m + 1) * c * sp.lpmv(m, v + 1, c))

    def t_int(mi, vi, mj, vj):
        return lambda t: 1 / np.sin(t)**3 * lap(mi, vi, t) * lap(mj, vj, t)
    return _assemble(model, z_int, t_int)


# sphharmlag.py:224-239
def eval_psi(model):
    def z_int(ki, kj):
        return lambda z: np.exp(-1 * z) * sp.eval_laguerre(ki, z) * sp.eval_laguerre(kj, z) * z**2

    def t_int(mi, vi, mj, vj):
        return lambda t: sp.lpmv(mi, vi, np.cos(t)) * sp.lpmv(mj, vj, np.cos(t)) * np.sin(t)
    return _assemble(model, z_int, t_int)


# ---- Psi by Gauss quadrature -----------------------------------------------------------------------
def eval_psi_gauss(model, ntheta=None):
    """Psi (sphharmlag.py:215-239) from closed-form / Gaussian quadrature instead of 3 x N(N+1)/2 adaptive QUADPACK calls.

    The three factors of psi_ij are convergent and smooth (unlike Omega's z-integral, which diverges - SURVEY F5 - so that
    the reference's Omega is a QUADPACK artefact no other quadrature can reproduce):
      z      int_0^inf e^-z L_ki L_kj z^2 dz   - a polynomial of degree <= 2 maxk against e^-z: Gauss-Laguerre with maxk + 2
             nodes is exact (finite MAX_Z_INT: the reference's adaptive quad call, once per (ki, kj));
      theta  int_0^cap P_vi^mi(cos t) P_vj^mj(cos t) sin t dt  - Gauss-Legendre on [0, cap], nodes growing with the degree;
      phi    int_0^2pi Az_i Az_j dphi = Kvm_i Kvm_j x (2 pi | pi | 0)  - analytic.
    One table per factor, the matrix is their outer combination: milliseconds at N = 144 (2.4 s with the de-duplicated
    QUADPACK calls of eval_psi, 10 s in the reference), 0.2 s at N = 1152 (52 s / ~25 min).  Agrees with the reference's
    own Psi to 1e-10 of max|Psi| (tests/test_host_tables.py) - QUADPACK's tolerance, not this routine's."""
    N, maxk, maxl = model.nbasis, model.maxk, model.maxl
    L2 = maxl**2
    klm = [model.basis_numbers(n) for n in range(L2)]
    nus = np.array([model.nu(n) for n in range(L2)], dtype=np.float64)
    ms = np.array([klm[n][2] for n in range(L2)], dtype=np.float64)
    # z
    if np.isinf(model.max_z_int):
        zx, zw = np.polynomial.laguerre.laggauss(maxk + 2)
        zw = zw * zx**2
        Lk = np.array([sp.eval_laguerre(k, zx) for k in range(maxk)])
        Iz = (Lk * zw) @ Lk.T
    else:
        # finite limit: e^-z z^2 L_ki L_kj on [0, Z] is no polynomial weight any more and a fixed Gauss-Legendre rule is only
        # converged while Z is small against its node count (a limit of 1e3 used as 'practically infinite' is not).  There
        # are only maxk (maxk + 1) / 2 such integrals: the reference's own adaptive quad call for each (sphharmlag.py:230).
        Iz = np.zeros((maxk, maxk))
        for ki in range(maxk):
            for kj in range(ki, maxk):
                Iz[ki, kj] = Iz[kj, ki] = scipy.integrate.quad(
                    lambda z: np.exp(-1 * z) * sp.eval_laguerre(ki, z) * sp.eval_laguerre(kj, z) * z**2, 0., model.max_z_int)[0]
    Iz = 0.5 * (Iz + Iz.T)                      # the products round asymmetrically; the reference fills Psi symmetrically
    # theta: signed order into lpmv, as the reference does
    if ntheta is None:
        ntheta = int(max(96, 4 * np.ceil(np.max(nus)) + 32))
    gx, gw = np.polynomial.legendre.leggauss(ntheta)
    t = 0.5 * model.cap_lim * (gx + 1.)
    tw = 0.5 * model.cap_lim * gw * np.sin(t)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Pm = np.array([sp.lpmv(ms[r], nus[r], np.cos(t)) for r in range(L2)])
    It = (Pm * tw) @ Pm.T
    It = 0.5 * (It + It.T)
    # phi: Az(v, m, p) = Kvm(v, |m|) * (sin(|m| p) if m < 0 else cos(|m| p))   (sphharmlag.py:263-281)
    K = np.array([model.Kvm(nus[r], abs(ms[r])) for r in range(L2)])
    same = ms[:, None] == ms[None, :]
    Ip = np.where(same, np.where(ms[:, None] == 0, 2. * np.pi, np.pi), 0.) * np.outer(K, K)
    ang = It * Ip
    kk = np.arange(N) // L2
    rr = np.arange(N) % L2
    return Iz[np.ix_(kk, kk)] * ang[np.ix_(rr, rr)]
