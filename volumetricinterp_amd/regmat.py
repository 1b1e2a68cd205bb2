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
