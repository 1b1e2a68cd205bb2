#!/usr/bin/env python3
"""CPU experiment: sweeps of round-robin (Brent-Luk) cyclic Jacobi on X(alpha) = AWA + alpha R at the default order,
for different initial index->slot arrangements.  Emulates the criterion of csrc/vi_jacobi.hip."""
import sys
import numpy as np

EPS = np.finfo(float).eps


def rr_pairs(order):
    """Generator of rounds: list of (p, q) index pairs; Brent-Luk music chairs on the given initial order."""
    n = len(order)
    m = n // 2
    top = list(order[:m])
    bot = list(order[m:])
    for _ in range(n - 1):
        yield list(zip(top, bot))
        ntop = [top[0], bot[0]] + top[1:m - 1]
        nbot = bot[1:] + [top[m - 1]]
        top, bot = ntop, nbot


def jacobi(X, order, floor, max_sweeps=40, sortrot=False):
    A = X.copy()
    n = A.shape[0]
    scale = np.max(np.abs(A))
    A /= scale
    hist = []
    for sweep in range(max_sweeps):
        nrot = 0
        for pairs in rr_pairs(order):
            p = np.array([a for a, b in pairs])
            q = np.array([b for a, b in pairs])
            app, aqq, apq = A[p, p], A[q, q], A[p, q]
            do = (np.abs(apq) > EPS * np.sqrt(np.abs(app * aqq))) & (np.abs(apq) > floor)
            nrot += int(do.sum())
            if not do.any():
                continue
            tau = (aqq - app) / (2 * np.where(do, apq, 1.))
            t = np.sign(tau) / (np.abs(tau) + np.sqrt(1 + tau * tau))
            t = np.where(tau == 0, 1., t)
            c = 1 / np.sqrt(1 + t * t)
            s = t * c
            c = np.where(do, c, 1.)
            s = np.where(do, s, 0.)
            if sortrot:
                # choose the rotation that leaves the larger diagonal in p (swap = extra quarter turn)
                npp = app - t * apq
                nqq = aqq + t * apq
                sw = do & (np.abs(npp) < np.abs(nqq))
                c, s = np.where(sw, -s, c), np.where(sw, c, s)
            J = np.eye(n)
            J[p, p] = c
            J[q, q] = c
            J[p, q] = s
            J[q, p] = -s
            A = J.T @ A @ J
            A = 0.5 * (A + A.T)
        off = np.sqrt(np.sum(A**2) - np.sum(np.diag(A)**2))
        hist.append((nrot, off))
        if nrot == 0:
            break
    return np.diag(A) * scale, hist


def main():
    f = np.load('tests/golden/fit_default.npz', allow_pickle=True)
    AWA, R = f['rec0_AWA'], f['R']
    n = AWA.shape[0]
    for la in [float(a) for a in sys.argv[1:]] or [-5., -20., -26.4, -40.]:
        X = AWA + 10.**la * R
        ref = np.sort(np.linalg.eigvalsh(X))
        d = np.abs(np.diag(X))
        orders = {
            'identity (top 0..m-1, bottom m..2m-1)': list(range(n)),
            'sorted desc': list(np.argsort(-d)),
            'sorted desc interleaved': None,
        }
        srt = list(np.argsort(-d))
        orders['sorted desc interleaved'] = srt[0::2] + srt[1::2]
        for name, order in orders.items():
            for sortrot in (False,):
                lam, hist = jacobi(X, order, 1e-22, sortrot=sortrot)
                err = np.max(np.abs(np.sort(lam) - ref) / np.maximum(np.abs(ref), 1e-16 * np.max(np.abs(ref))))
                print('log10a=%6.1f  %-40s sortrot=%d sweeps=%2d  rot/sweep=%s  err=%.1e' %
                      (la, name, sortrot, len(hist), [h[0] for h in hist], err))


if __name__ == '__main__':
    main()
