"""Oracle (test infrastructure): an ACCURATE evaluation of the definition behind the reference's solve.

``scipy.linalg.lstsq(X, y)`` at ``volumetricinterp/interpolate.py:462`` means: the minimum-norm solution of X c = y with
the singular values below eps * sigma_max dropped.  At the default order X(alpha) has eigenvalues at that threshold and
LAPACK resolves them to 10-100 % (tools/gen_exact.py), so LAPACK cannot arbitrate between two answers that differ in
the third digit.  This module evaluates the same definition for a symmetric X in plain NumPy with relative accuracy on
the small eigenvalues: one column-pivoted QR similarity step (X1 = Q^T X Q, SciPy's Householder QR) followed by a
cyclic Jacobi eigen-iteration with the relative rotation criterion |a_pq| <= eps sqrt(|a_pp a_qq|) - the textbook
algorithm (Demmel & Veselic: Jacobi's method is more accurate than QR), not the GPU kernel's schedule.  It is checked
against 50-digit arithmetic in tests/test_oracle_golden.py (rank, chi^2 and A c of tests/golden/exact_default_c2.npz)
and used by the GPU parity tests as an independent CPU arbiter.  Seconds per solve; never part of the product.
"""
import numpy as np
import scipy.linalg

EPS = np.finfo(float).eps


def _round_robin(n):
    """n - 1 rounds of n / 2 disjoint index pairs (circle method), n even."""
    idx = list(range(n))
    out = []
    for _ in range(n - 1):
        out.append((np.array([idx[i] for i in range(n // 2)]), np.array([idx[n - 1 - i] for i in range(n // 2)])))
        idx = [idx[0]] + [idx[-1]] + idx[1:-1]
    return out


def jacobi_eigh(X, max_sweeps=60):
    """Eigen-decomposition of a symmetric matrix by cyclic Jacobi (disjoint pairs rotated together, round robin).
    Returns (lam, V, sweeps) with X = V diag(lam) V^T; rotations use the relative criterion."""
    N = X.shape[0]
    n = N + (N & 1)
    A = np.zeros((n, n))
    A[:N, :N] = 0.5 * (X + X.T)
    V = np.eye(n)
    rounds = _round_robin(n)
    for sweep in range(max_sweeps):
        nrot = 0
        for P, Q in rounds:
            app, aqq, apq = A[P, P], A[Q, Q], A[P, Q]
            rot = (apq * apq > EPS * EPS * np.abs(app * aqq)) & (apq != 0.0)
            if not rot.any():
                continue
            d = aqq - app
            with np.errstate(all='ignore'):
                t = np.copysign(2.0 * apq, d * apq) / (np.abs(d) + np.sqrt(d * d + 4.0 * apq * apq))
                t = np.where(d == 0.0, np.sign(apq), t)
                c = 1.0 / np.sqrt(t * t + 1.0)
                s = t * c
            c = np.where(rot, c, 1.0)
            s = np.where(rot, s, 0.0)
            nrot += int(rot.sum())
            rp, rq = A[P].copy(), A[Q].copy()
            A[P], A[Q] = c[:, None] * rp - s[:, None] * rq, s[:, None] * rp + c[:, None] * rq
            cp, cq = A[:, P].copy(), A[:, Q].copy()
            A[:, P], A[:, Q] = c * cp - s * cq, s * cp + c * cq
            A[P[rot], Q[rot]] = 0.0
            A[Q[rot], P[rot]] = 0.0
            vp, vq = V[:, P].copy(), V[:, Q].copy()
            V[:, P], V[:, Q] = c * vp - s * vq, s * vp + c * vq
        if nrot == 0:
            break
    return np.diag(A)[:N].copy(), V[:N, :N], sweep + 1


def lstsq_accurate(X, y, rcond=EPS):
    """Minimum-norm solution of the symmetric system X c = y with |lambda| <= rcond * max|lambda| dropped; (c, rank)."""
    X = np.asarray(X, dtype=np.float64)
    mx = np.max(np.abs(X))
    sc = 2.0**(1 - np.frexp(mx)[1]) if mx > 0 else 1.0
    Xs = X * sc
    Q, _, _ = scipy.linalg.qr(Xs, pivoting=True)
    X1 = Q.T @ Xs @ Q
    lam, V, _ = jacobi_eigh(0.5 * (X1 + X1.T))
    keep = np.abs(lam) > rcond * np.max(np.abs(lam))
    g = np.where(keep, (V.T @ (Q.T @ y)) / np.where(keep, lam, 1.0), 0.0)
    return (Q @ (V @ g)) * sc, int(keep.sum())


def chi2_accurate(A, b, W, R, alpha):
    """chi^2 of the regularised fit at alpha (interpolate.py:255-259 with the solve of :462 evaluated accurately)."""
    X = np.einsum('ji,j,jk->ik', A, W, A) + alpha * R
    y = np.einsum('ji,j,j->i', A, W, b)
    c, rank = lstsq_accurate(X, y)
    return float(np.sum((A @ c - b)**2 * W)), c, rank
