"""Oracle (test infrastructure): evaluation of a fitted model.

Restates ``volumetricinterp/estimate.py`` of the reference: ``__call__`` :75-123,
``check_hull`` :153-178, ``get_C`` :180-221.
"""
import datetime as dt

import numpy as np
from scipy.spatial import ConvexHull

from .geodesy import geodetic2ecef


def get_C(t, time, Coeffs, Covariance, timetol=60., timeinterp=False):
    """estimate.py:180-221.  ``t`` is a naive-UTC datetime."""
    t0 = (t - dt.datetime(1970, 1, 1)).total_seconds()
    mt = np.mean(time, axis=1)
    try:
        if timeinterp:
            i = np.argwhere((t0 >= mt[:-1]) & (t0 < mt[1:])).flatten()[0]
            T = (t0 - mt[i]) / (mt[i + 1] - mt[i])
            C = (1 - T) * Coeffs[i, :] + T * Coeffs[i + 1, :]
            dC = (1 - T) * Covariance[i, :, :] + T * Covariance[i + 1, :, :]
        else:
            i = np.argmin(np.abs(mt - t0))
            if np.abs(mt[i] - t0) > timetol:
                raise IndexError
            C = Coeffs[i]
            dC = Covariance[i]
    except IndexError:
        raise ValueError('Requested time out of range of data file.')
    return C, dC


def check_hull(hull_vert, lat0, lon0, alt0):
    """estimate.py:153-178: one Qhull per query point (slow, faithful)."""
    lat0, lon0, alt0 = (np.asarray(a, dtype=np.float64) for a in (lat0, lon0, alt0))
    hull = ConvexHull(hull_vert)
    out = []
    for lat, lon, alt in zip(lat0.ravel(), lon0.ravel(), alt0.ravel()):
        x, y, z = geodetic2ecef(lat, lon, alt)
        pnts = np.append(hull_vert, np.array([[x, y, z]]), axis=0)
        out.append(bool(np.array_equal(hull.vertices, ConvexHull(pnts).vertices)))
    return np.array(out).reshape(alt0.shape)


def evaluate(model, C, gdlat, gdlon, gdalt, hull_vert=None):
    """estimate.py:110-123 (calcgrad/calcerr are dead code in the reference)."""
    A = model.basis(gdlat, gdlon, gdalt)
    parameter = np.einsum('...i,i->...', A, C)
    if hull_vert is not None:
        parameter[~check_hull(hull_vert, gdlat, gdlon, gdalt)] = np.nan
    return parameter


def evaluate_gradient(model, C, gdlat, gdlon, gdalt, hull_vert=None):
    """Gradient of the fitted parameter: grad_basis (sphharmlag.py:148-184) contracted with C - what the dead code after
    the return of Estimate.__call__ (estimate.py:125-147) was meant to deliver.  Shape gdlat.shape + (3,)."""
    gdlat = np.asarray(gdlat, dtype=np.float64)
    G = model.grad_basis(gdlat.ravel(), np.asarray(gdlon, dtype=np.float64).ravel(),
                         np.asarray(gdalt, dtype=np.float64).ravel())                  # (P, 3, N)
    out = np.einsum('pcn,n->pc', G, np.asarray(C, dtype=np.float64))
    if hull_vert is not None:
        out[~check_hull(hull_vert, gdlat.ravel(), np.asarray(gdlon).ravel(), np.asarray(gdalt).ravel())] = np.nan
    return out.reshape(gdlat.shape + (3,))


def evaluate_error(model, dC, gdlat, gdlon, gdalt, hull_vert=None):
    """Standard error of the fitted parameter by first-order propagation of the coefficient covariance:
    sqrt(a^T dC a) per point (the `calcerr` output of the dead branch, estimate.py:139-145).  Shape of gdlat."""
    gdlat = np.asarray(gdlat, dtype=np.float64)
    A = model.basis(gdlat.ravel(), np.asarray(gdlon, dtype=np.float64).ravel(), np.asarray(gdalt, dtype=np.float64).ravel())
    with np.errstate(invalid='ignore'):
        out = np.sqrt(np.einsum('pn,nm,pm->p', A, np.asarray(dC, dtype=np.float64), A))
    if hull_vert is not None:
        out[~check_hull(hull_vert, gdlat.ravel(), np.asarray(gdlon).ravel(), np.asarray(gdalt).ravel())] = np.nan
    return out.reshape(gdlat.shape)
