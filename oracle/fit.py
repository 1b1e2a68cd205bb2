"""Oracle (test infrastructure): regularised weighted least-squares fit.

Restates the fit core of ``volumetricinterp/interpolate.py`` of the reference:
``eval_C`` :432-469, ``chi2objfunct`` :220-261, ``chi2`` :152-218,
``find_reg_param`` :97-147, the record loop of ``calc_coeffs`` :511-579 and
``compute_hull`` :409-426.  Algorithmic structure is kept faithful on purpose
(it is the CPU baseline): A^T W A is rebuilt with an un-optimised three-operand
einsum for every regularisation parameter tried, and the chi^2(alpha) walk is
repeated for every scale factor.
"""
import numpy as np
import scipy.linalg
import scipy.optimize
from scipy.spatial import ConvexHull

from .geodesy import geodetic2ecef

SCALE_FACTORS = (0.6, 0.7, 0.8, 0.9, 1.0)      # interpolate.py:173


def eval_C(A, b, W, reg_matrices, reg_params, regularization_list, calccov=False, lapack_driver=None):
    """interpolate.py:432-469.  lapack_driver: None = SciPy's default (gelsd), what the reference runs; 'gelss' / 'gelsy' =
    the other LAPACK routines scipy.linalg.lstsq offers for the same definition (tests measure how much of the reference's
    answer is its library routine)."""
    AWA = np.einsum('ji,j,jk->ik', A, W, A)
    X = AWA.copy()
    y = np.einsum('ji,j,j->i', A, W, b)
    for reg in regularization_list:
        X = X + reg_params[reg] * reg_matrices[reg]
    # LAPACK gelsd, cond=None -> eps; check_finite=True raises ValueError on NaN/inf
    C = np.squeeze(scipy.linalg.lstsq(X, y, lapack_driver=lapack_driver)[0])
    if calccov:
        H = scipy.linalg.pinv(X)
        dC = np.einsum('ij,jk,kl->il', H, AWA, H)
        return C, dC
    return C


def chi2objfunct(alpha, A, b, W, reg_matrices, nu, reg, regularization_list, counter=None):
    """interpolate.py:220-261: chi^2(10**alpha) - nu with all other parameters 0."""
    reg_params = {rl: (np.power(10., alpha) if rl == reg else 0.) for rl in regularization_list}
    C = eval_C(A, b, W, reg_matrices, reg_params, regularization_list)
    if counter is not None:
        counter[0] += 1
    val = np.einsum('ji,i->j', A, C)
    return sum((val - b)**2 * W) - nu


def chi2_search(A, b, W, reg_matrices, reg, regularization_list, counter=None, trace=None):
    """interpolate.py:152-218.  Returns alpha (0 for 'too smooth'), raises
    ValueError when no scale factor brackets a root in (1e-100, 1)."""
    N = len(b)
    bracket = False
    args = None
    for sf in SCALE_FACTORS:
        nu = N * sf
        args = (A, b, W, reg_matrices, nu, reg, regularization_list, counter)
        alpha0, val0, alpha = 0., 1., 0.
        val = chi2objfunct(alpha, *args)
        if val < 0:
            if trace is not None:
                trace.update(sf=sf, outcome='too_smooth')
            return 0
        while val0 * val > 0:
            bracket = True
            val0 = val
            alpha0 = alpha
            alpha = alpha - 1.
            val = chi2objfunct(alpha, *args)
            if alpha < -100.:
                bracket = False
                break
        if bracket:
            break
    if not bracket:
        if trace is not None:
            trace.update(sf=None, outcome='no_root')
        raise ValueError('Could not find any roots to the objective function chi^2-nu in the range (1e-100,1).')
    solution = scipy.optimize.brentq(chi2objfunct, alpha, alpha0, args=args, disp=True)
    if trace is not None:
        trace.update(sf=sf, outcome='root', bracket=(alpha, alpha0), log10_alpha=solution)
    return np.power(10., solution)


def gcvobjfunct(alpha, A0, b0, W0, reg_matrices, reg, regularization_list, counter=None):
    """interpolate.py:299-351: leave-one-out sum of squared weighted residuals at 10**alpha."""
    alpha = float(np.squeeze(alpha))
    reg_params = {rl: (np.power(10., alpha) if rl == reg else 0.) for rl in regularization_list}
    residuals = []
    for i in range(len(b0)):
        Ai, bi, Wi = A0[i, :], b0[i], W0[i]
        A = np.delete(A0, i, 0)
        b = np.delete(b0, i, 0)
        W = np.delete(W0, i, 0)
        C = eval_C(A, b, W, reg_matrices, reg_params, regularization_list)
        if counter is not None:
            counter[0] += 1
        val = np.squeeze(np.dot(Ai, C))
        residuals.append((val - bi)**2 * Wi)
    return sum(residuals)


def gcv_search(A, b, W, reg_matrices, reg, regularization_list, counter=None):
    """interpolate.py:263-297: Nelder-Mead from alpha0 = -20."""
    sol = scipy.optimize.minimize(gcvobjfunct, -20., args=(A, b, W, reg_matrices, reg, regularization_list, counter),
                                  method='Nelder-Mead')
    if not sol.success:
        raise ValueError('Minima of GCV function could not be found')
    return np.power(10., sol.x[0])


def find_reg_param(A, b, W, reg_matrices, regularization_list, counter=None, traces=None, method='chi2'):
    """interpolate.py:97-147; method 'chi2' is the default (:135-136), 'gcv' the other working one."""
    out = {}
    for rl in regularization_list:
        tr = {} if traces is not None else None
        try:
            if method == 'gcv':
                out[rl] = gcv_search(A, b, W, reg_matrices, rl, regularization_list, counter)
                continue
            out[rl] = chi2_search(A, b, W, reg_matrices, rl, regularization_list, counter, tr)
        except ValueError:
            out[rl] = np.nan
        if traces is not None:
            traces[rl] = tr
    return out


def compute_hull_vertices(lat, lon, alt):
    """interpolate.py:409-426."""
    x, y, z = geodetic2ecef(lat, lon, alt)
    R = np.array([x, y, z]).T
    return R[ConvexHull(R).vertices]


def fit_records(model, lat, lon, alt, value, error, reg_matrices, regularization_list, counter=None, method='chi2'):
    """Record loop of calc_coeffs, interpolate.py:511-579 (no file I/O).

    value, error: (T, P).  Returns Coeffs (T,N), Covariance (T,N,N), chi_sq (T,),
    reg_params list (one dict per record).
    """
    Coeffs, Covariance, chi_sq, params = [], [], [], []
    N = model.nbasis
    for ne0, er0 in zip(value, error):
        fin = np.isfinite(ne0)                       # mask derives from the value only (:516-520)
        lat0, lon0, alt0 = lat[fin], lon[fin], alt[fin]
        er0 = er0[fin]
        ne0 = ne0[fin]
        W = np.array(er0**(-2))
        b = ne0
        A = model.basis(lat0, lon0, alt0)
        reg_params = find_reg_param(A, b, W, reg_matrices, regularization_list, counter, method=method)
        params.append(reg_params)
        if np.any(np.isnan([v for v in reg_params.values()])):
            Coeffs.append(np.full(N, np.nan))
            Covariance.append(np.full((N, N), np.nan))
            chi_sq.append(np.nan)
            continue
        C, dC = eval_C(A, b, W, reg_matrices, reg_params, regularization_list, calccov=True)
        c2 = sum((np.squeeze(np.dot(A, C)) - np.squeeze(b))**2 * np.squeeze(W))
        Coeffs.append(C)
        Covariance.append(dC)
        chi_sq.append(c2)
    return np.array(Coeffs), np.array(Covariance), np.array(chi_sq), params
