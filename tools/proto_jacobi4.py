"""CPU prototype of the rotation ordering of k_jacobi_solve (csrc/vi_jacobi.hip): "matches" of four indices.

The N indices sit in slots; unit u = slots (2u, 2u+1); match a = units (2a, 2a+1) = slots 4a..4a+3 = (U0, U1, V0, V1).
One OUTER round rotates, in every match, the four cross pairs of its two units - inner round 1: (U0,V0), (U1,V1), inner
round 2: (U0,V1), (U1,V0) - and then moves the UNITS one step along the Brent-Luk ring, so after m - 1 outer rounds
every pair of units has met once.  The intra-unit pairs (U0,U1), (V0,V1) get one extra round per sweep.  Every pair of
indices is rotated exactly once per sweep, as in the classical round-robin ordering, but the matrix passes through the
registers half as often.  This script checks the ordering (eigenvalues, truncated solve by reverse replay of the
rotation log) against LAPACK and compares its sweep count with the one-pair-per-round ordering.
Usage: python tools/proto_jacobi4.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPS = np.finfo(float).eps


def slot_next(s, m):            # Brent-Luk on 2m slots, slot 0 fixed
    if s == 0:
        return 0
    if s == 1:
        return 2
    if s & 1:
        return s - 2
    return 2 * m - 1 if s == 2 * m - 2 else s + 2


def rot_params(app, aqq, apq, drop, floor):
    aa = abs(apq)
    tiny = max(abs(app), abs(aqq), aa) < 0.0625 * drop
    if aa * aa > EPS * EPS * abs(app * aqq) and aa > floor and not tiny:
        d = aqq - app
        t = np.copysign(2.0 * apq, d * apq) / (abs(d) + np.sqrt(d * d + 4.0 * apq * apq))
        c = 1.0 / np.sqrt(t * t + 1.0)
        return c, t * c, True
    return 1.0, 0.0, False


def apply_rot(A, y, p, q, c, s):
    """A <- J^T A J, y <- J^T y for the plane rotation J in (p, q)."""
    rp, rq = A[p].copy(), A[q].copy()
    A[p], A[q] = c * rp - s * rq, s * rp + c * rq
    cp, cq = A[:, p].copy(), A[:, q].copy()
    A[:, p], A[:, q] = c * cp - s * cq, s * cp + c * cq
    yp, yq = y[p], y[q]
    y[p], y[q] = c * yp - s * yq, s * yp + c * yq


def converged(A, drop, floor):
    d = np.abs(np.diag(A))
    off = np.abs(A - np.diag(np.diag(A)))
    big = np.maximum(np.maximum(d[:, None], d[None, :]), off)
    viol = (off * off > EPS * EPS * np.abs(np.outer(np.diag(A), np.diag(A)))) & (off > floor) & ~(big < 0.0625 * drop)
    return not viol.any()


def jacobi4(X, yv, rcond=EPS, floor=1e-22, max_sweeps=40):
    N = X.shape[0]
    Np = (N + 3) & ~3
    m, M = Np // 2, Np // 4
    A = np.zeros((Np, Np))
    A[:N, :N] = X
    y = np.zeros(Np)
    y[:N] = yv
    drop = rcond * np.max(np.abs(np.diag(A)))
    log = []                                     # per round: (kind, [(c, s) x 4 per match])
    perm = np.array([2 * slot_next(s >> 1, M) + (s & 1) for s in range(Np)])      # unit permutation on slots
    sweeps = 0
    for sweep in range(max_sweeps):
        rotated = False
        for r in range(m):                       # round 0: intra-unit pairs; rounds 1 .. m-1: cross pairs + permutation
            rots = np.tile(np.array([1.0, 0.0]), (M, 4, 1))
            inner = [((0, 1), (2, 3))] if r == 0 else [((0, 2), (1, 3)), ((0, 3), (1, 2))]
            for k, pairs in enumerate(inner):
                # parallel round: all parameters from the matrix at the start of the inner round
                todo = []
                for a in range(M):
                    for j, (p, q) in enumerate(pairs):
                        P, Q = 4 * a + p, 4 * a + q
                        c, s, did = rot_params(A[P, P], A[Q, Q], A[P, Q], drop, floor)
                        rots[a, 2 * k + j] = (c, s)
                        rotated |= did
                        todo.append((P, Q, c, s))
                for P, Q, c, s in todo:
                    apply_rot(A, y, P, Q, c, s)
                    A[P, Q] = A[Q, P] = 0.0 if (c, s) != (1.0, 0.0) else A[P, Q]
            log.append((r == 0, rots))
            if r > 0:
                A2 = np.empty_like(A)
                A2[np.ix_(perm, perm)] = A
                A = A2
                y2 = np.empty_like(y)
                y2[perm] = y
                y = y2
        sweeps += 1
        if not rotated or converged(A, drop, floor):
            break
    lam = np.diag(A).copy()
    keep = np.abs(lam) > rcond * np.max(np.abs(lam))
    g = np.where(keep, y / np.where(keep, lam, 1.0), 0.0)
    # reverse replay: undo (permutation, inner 2, inner 1) round by round
    for intra, rots in reversed(log):
        if not intra:
            g = g[perm]                          # old[s] = new[perm[s]]
            order = [((0, 3), (1, 2), 2), ((0, 2), (1, 3), 0)]
        else:
            order = [((0, 1), (2, 3), 0)]
        for p1, p2, base in order:
            for a in range(M):
                for j, (p, q) in enumerate((p1, p2)):
                    c, s = rots[a, base + j]
                    P, Q = 4 * a + p, 4 * a + q
                    gp, gq = g[P], g[Q]
                    g[P], g[Q] = c * gp + s * gq, -s * gp + c * gq
    return lam[:N] if Np == N else lam, g[:N], int(keep.sum()), sweeps


def jacobi2(X, yv, rcond=EPS, floor=1e-22, max_sweeps=40):
    """The one-pair-per-round round-robin ordering (sweep count only)."""
    N = X.shape[0]
    Np = (N + 1) & ~1
    m = Np // 2
    A = np.zeros((Np, Np))
    A[:N, :N] = X
    y = np.zeros(Np)
    drop = rcond * np.max(np.abs(np.diag(A)))
    perm = np.array([slot_next(s, m) for s in range(Np)])
    for sweep in range(max_sweeps):
        rotated = False
        for r in range(2 * m - 1):
            todo = []
            for P in range(m):
                c, s, did = rot_params(A[2 * P, 2 * P], A[2 * P + 1, 2 * P + 1], A[2 * P, 2 * P + 1], drop, floor)
                rotated |= did
                todo.append((2 * P, 2 * P + 1, c, s))
            for P, Q, c, s in todo:
                apply_rot(A, y, P, Q, c, s)
            A2 = np.empty_like(A)
            A2[np.ix_(perm, perm)] = A
            A = A2
        if not rotated or converged(A, drop, floor):
            return sweep + 1
    return max_sweeps


def scaled(X):
    mx = np.max(np.abs(X))
    return X * 2.0**(1 - np.frexp(mx)[1])


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    for N in (8, 10, 32, 37):
        Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
        lam = rng.standard_normal(N) * 10.0**rng.uniform(-8, 0, N)
        X = (Q * lam) @ Q.T
        X = 0.5 * (X + X.T)
        yv = rng.standard_normal(N)
        l4, C, rk, sw = jacobi4(X, yv)
        ref = np.linalg.lstsq(X, yv, rcond=EPS)[0]
        print('N=%d sweeps %d rank %d  eig err %.1e  rel(C) %.1e' % (
            N, sw, rk, np.max(np.abs(np.sort(l4[:N]) - np.sort(np.linalg.eigvalsh(X)))) / np.max(np.abs(lam)),
            np.linalg.norm(C - ref) / np.linalg.norm(ref)))
    for name in ('fit_k8l2', 'fit_default'):
        f = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'), allow_pickle=True)
        X, yv = f['rec0_X'], f['rec0_y']
        Xs = scaled(X)
        sc = X.flat[0] / Xs.flat[0]
        l4, C, rk, sw = jacobi4(Xs, yv)
        C = C / sc
        ref = f['Coeffs'][0]
        print('%s N=%d: matches-of-4 ordering %d sweeps (rank %d, rel(C) vs reference %.2e); pair ordering %d sweeps'
              % (name, X.shape[0], sw, rk, np.linalg.norm(C - ref) / np.linalg.norm(ref), jacobi2(Xs, yv)))
