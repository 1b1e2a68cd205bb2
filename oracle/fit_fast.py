"""Oracle (test infrastructure / CPU baseline only): the "optimised CPU" variant of BASELINE.md section 4.

Same alpha and coefficients as ``oracle.fit``, bit for bit (same calls into LAPACK on the same matrices, same
contractions inside chi^2, same Brent iteration; the covariance product uses matmul and agrees to rounding), with the two
redundancies of the reference (volumetricinterp/interpolate.py:180-214, :255, :456) removed: A^T W A and A^T W b are
formed once per record instead of once per trial alpha, and chi^2(alpha) is memoised, so the bracket walk is not
repeated for every scale factor.  Reported beside the faithful variant for context; never a parity yardstick.
"""
import numpy as np
import scipy.linalg
import scipy.optimize

SCALE_FACTORS = (0.6, 0.7, 0.8, 0.9, 1.0)      # interpolate.py:173


def fit_record(A, b, W, R, counter=None):
    """One record, one regularisation matrix, method chi2, with covariance.  Returns C, dC, chi2, alpha
    (NaN row / alpha = 0 conventions of interpolate.py:142-145, :189-191)."""
    N = A.shape[1]
    AWA = np.einsum('ji,j,jk->ik', A, W, A)          # the reference's own (unoptimised) contraction, once
    y = np.einsum('ji,j,j->i', A, W, b)
    memo = {}

    def chi2(a):
        if a not in memo:
            C = np.squeeze(scipy.linalg.lstsq(AWA + np.power(10., a) * R, y)[0])
            if counter is not None:
                counter[0] += 1
            memo[a] = sum((np.einsum('ji,i->j', A, C) - b)**2 * W)        # the reference's own contraction
        return memo[a]

    npts = len(b)
    bracket = False
    alpha = alpha0 = 0.
    nu = 0.
    for sf in SCALE_FACTORS:
        nu = npts * sf
        alpha0, val0, alpha = 0., 1., 0.
        val = chi2(alpha) - nu
        if val < 0:
            alpha_out = 0.
            break
        while val0 * val > 0:
            bracket = True
            val0, alpha0 = val, alpha
            alpha = alpha - 1.
            val = chi2(alpha) - nu
            if alpha < -100.:
                bracket = False
                break
        if bracket:
            alpha_out = np.power(10., scipy.optimize.brentq(lambda a: chi2(a) - nu, alpha, alpha0))
            break
    else:
        return np.full(N, np.nan), np.full((N, N), np.nan), np.nan, np.nan
    X = AWA + alpha_out * R
    C = np.squeeze(scipy.linalg.lstsq(X, y)[0])
    H = scipy.linalg.pinv(X)
    dC = H @ AWA @ H
    return C, dC, sum((A @ C - b)**2 * W), alpha_out
