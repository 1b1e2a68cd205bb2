#!/usr/bin/env python3
"""Exact-arithmetic answers for the truncated solve at the default order (build container only; needs mpmath).

The reference solves X c = y by scipy.linalg.lstsq (LAPACK gelsd, rcond = eps): mathematically the minimum-norm
solution with the singular values below eps * sigma_max dropped.  LAPACK evaluates that definition with absolute errors of
eps * sigma_max on every singular value, i.e. 10-100 % on the ones near the cut, and its answer moves by O(1) under one
ulp on alpha; this script evaluates the SAME definition in 50-digit arithmetic (mpmath symmetric eigen-decomposition of
the float64 matrix X), which gives the well-defined answer both LAPACK and the GPU solver approximate.  Systems: the
reference's own X = A^T W A + alpha R and y of fixture fit_default_c2 (26 x 100, N = 144), at the reference's alpha of
each record and 0.2 decades below.  Output tests/golden/exact_default_c2.npz (data only)."""
import math
import os
import sys

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                         # noqa: E402  (the CPU restatement: basis and normal equations)

mp.mp.dps = 50
EPS = np.finfo(float).eps
f = np.load(os.path.join(ROOT, 'tests', 'golden', 'fit_default_c2.npz'), allow_pickle=True)
o = oracle.SphHarmLagOracle()
A = o.basis(f['lat'], f['lon'], f['alt'])
R = f['R']
Xs, ys, Cs, chis, ranks, las, recs, cuts = [], [], [], [], [], [], [], []
for t in range(f['value'].shape[0]):
    if not np.isfinite(f['alpha'][t]) or f['alpha'][t] <= 0:
        continue
    b, W = f['value'][t], f['error'][t]**-2.
    AWA = np.einsum('ji,j,jk->ik', A, W, A)
    y = np.einsum('ji,j,j->i', A, W, b)
    for dl in (0.0, -0.2):
        la = math.log10(f['alpha'][t]) + dl
        X = AWA + 10.**la * R
        E, Q = mp.eigsy(mp.matrix(X.tolist()))
        lam = np.array([float(x) for x in E])
        thr = EPS * np.max(np.abs(lam))
        ym = [mp.mpf(float(v)) for v in y]
        C = [mp.mpf(0)] * len(y)
        kept = 0
        for i in range(len(lam)):
            if abs(lam[i]) > thr:
                kept += 1
                g = sum(Q[r, i] * ym[r] for r in range(len(y))) / E[i]
                for r in range(len(y)):
                    C[r] += Q[r, i] * g
        Cn = np.array([float(c) for c in C])
        srt = np.sort(np.abs(lam))[::-1]
        Xs.append(X); ys.append(y); Cs.append(Cn); ranks.append(kept); las.append(la); recs.append(t)
        chis.append(float(sum((A @ Cn - b)**2 * W)))
        cuts.append(srt[kept - 2:kept + 2] / srt[0])
        print('record %d log10 alpha %.5f: rank %d chi2 %.6f  |lambda|/max around the cut %s' % (t, la, kept, chis[-1], cuts[-1]))
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'exact_default_c2.npz'), X=np.array(Xs), y=np.array(ys),
                    C=np.array(Cs), chi2=np.array(chis), rank=np.array(ranks), log10_alpha=np.array(las),
                    record=np.array(recs), around_cut=np.array(cuts))
