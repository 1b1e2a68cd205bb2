"""Regularisation-parameter search (chi^2 = nu), batched over records.

Host-side scalar logic of the reference's ``Interpolate.chi2`` / ``chi2objfunct``
(volumetricinterp/interpolate.py:152-261), restated as coroutines so that the
expensive part - one regularised solve + chi^2 per requested alpha - is served
in *batches* by the GPU: every record runs the reference's control flow
literally (scale factors .6 .. 1., bracket walk alpha = 0, -1, ... -101, then
Brent), but whenever it needs chi^2(alpha) it yields, and the driver gathers the
requests of all records into one ``vi_form_system`` / ``vi_solve_trunc`` /
``vi_chi2`` launch.  chi^2(alpha) is memoised per record (the reference
recomputes the identical table for each scale factor and brentq re-evaluates
both bracket ends), and the walk is prefetched a few steps ahead to cut GPU
round trips; neither changes any value the control flow sees.

``brentq_gen`` restates ``scipy.optimize.brentq`` (SciPy's Zeros/brentq.c,
called at interpolate.py:214 with the defaults xtol = 2e-12, rtol = 4 eps,
maxiter = 100).
"""
import math

import numpy as np

SCALE_FACTORS = (0.6, 0.7, 0.8, 0.9, 1.0)      # interpolate.py:173
XTOL = 2e-12
RTOL = 4 * np.finfo(np.float64).eps
MAXITER = 100

NO_ROOT_MSG = 'Could not find any roots to the objective function chi^2-nu in the range (1e-100,1).'


def _signbit(x):
    return math.copysign(1.0, x) < 0


def brentq_gen(xa, xb, fa=None, fb=None, xtol=XTOL, rtol=RTOL, maxiter=MAXITER):
    """Coroutine form of brentq: yields x, receives f(x); returns (root, iterations, funcalls, other_end) with other_end
    the far end of the final bracket (None when an end point is an exact zero).

    fa / fb may be supplied when already known (the function is deterministic)."""
    xpre, xcur = xa, xb
    xblk, fblk, spre, scur = 0., 0., 0., 0.
    funcalls = 0
    if fa is None:
        fa = yield xpre
        funcalls += 1
    if fb is None:
        fb = yield xcur
        funcalls += 1
    fpre, fcur = fa, fb
    if fpre == 0:
        return xpre, 0, funcalls, None
    if fcur == 0:
        return xcur, 0, funcalls, None
    if _signbit(fpre) == _signbit(fcur):
        raise ValueError('f(a) and f(b) must have different signs')
    for it in range(1, maxiter + 1):
        if fpre != 0 and fcur != 0 and (_signbit(fpre) != _signbit(fcur)):
            xblk, fblk = xpre, fpre
            spre = scur = xcur - xpre
        if abs(fblk) < abs(fcur):
            xpre, xcur, xblk = xcur, xblk, xcur
            fpre, fcur, fblk = fcur, fblk, fcur
        delta = (xtol + rtol * abs(xcur)) / 2
        sbis = (xblk - xcur) / 2
        if fcur == 0 or abs(sbis) < delta:
            return xcur, it, funcalls, xblk
        if abs(spre) > delta and abs(fcur) < abs(fpre):
            if xpre == xblk:
                stry = -fcur * (xcur - xpre) / (fcur - fpre)                      # secant
            else:
                dpre = (fpre - fcur) / (xpre - xcur)                              # inverse quadratic
                dblk = (fblk - fcur) / (xblk - xcur)
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre))
            if 2 * abs(stry) < min(abs(spre), 3 * abs(sbis) - delta):
                spre, scur = scur, stry
            else:
                spre, scur = sbis, sbis
        else:
            spre, scur = sbis, sbis
        xpre, fpre = xcur, fcur
        if abs(scur) > delta:
            xcur += scur
        else:
            xcur += delta if sbis > 0 else -delta
        fcur = yield xcur
        funcalls += 1
    raise RuntimeError('Failed to converge after %d iterations.' % maxiter)


MS_XTOL = 1e-7          # multisection stops at this bracket width (log10 alpha); the secant point finishes


def multisection_gen(xa, xb, fa, fb, K, xtol=MS_XTOL):
    """Latency-optimised replacement for the sequential Brent iteration when the GPU is otherwise idle (a
    single record or a few): every round evaluates K equispaced interior points of the current bracket *in
    one batch* and keeps the sub-interval with the sign change, shrinking the bracket by K+1 per round until it
    is shorter than xtol; the root is then the secant point of the last bracket (error ~ xtol^2 f''/f').

    GUARD.  chi^2(alpha) - nu is not monotone (the curvature matrix is indefinite) and a unit bracket can hold
    several roots; which of them ``brentq`` returns is decided by its iterate sequence.  Multisection is
    therefore only used while every round shows EXACTLY ONE sign change among its K+2 finite samples - then the
    bracket holds one root at the sampling resolution and any bracketing method, Brent included, converges to
    it.  On anything else (a second sign change, a NaN, an exact zero) the generator returns None and the
    caller runs the reference's Brent iteration on the original bracket.

    Yields a tuple of K abscissae, receives the K function values.  Returns (root, rounds, funcalls) or None."""
    lo, hi, flo, fhi = xa, xb, fa, fb
    if flo == 0 or fhi == 0 or math.isnan(flo) or math.isnan(fhi) or _signbit(flo) == _signbit(fhi):
        return None
    rounds = funcalls = 0
    while abs(hi - lo) > xtol:
        xs = tuple(lo + (hi - lo) * (k + 1) / (K + 1) for k in range(K))
        fs = yield xs
        rounds += 1
        funcalls += K
        pts = [(lo, flo)] + list(zip(xs, fs)) + [(hi, fhi)]
        if any(math.isnan(f) or f == 0 for _, f in pts):
            return None
        changes = [j for j in range(1, len(pts)) if _signbit(pts[j - 1][1]) != _signbit(pts[j][1])]
        if len(changes) != 1:
            return None
        j = changes[0]
        (lo, flo), (hi, fhi) = pts[j - 1], pts[j]
        if rounds > 60:
            return None
    if fhi == flo:
        return None
    return lo - flo * (hi - lo) / (fhi - flo), rounds, funcalls


def chi2_search_gen(npts, multisection=0):
    """Coroutine form of Interpolate.chi2 (interpolate.py:152-218) for one record.

    Yields log10(alpha) (or a tuple of them), receives chi^2 at that alpha (or a list).  Returns
    (outcome, alpha, info) with outcome in {'too_smooth', 'no_root', 'root'}; alpha is 0, NaN or 10**root as
    in the reference.  multisection = K > 0: try the guarded K-point multisection first (see multisection_gen),
    falling back to the Brent iteration when the bracket is not provably single-rooted."""
    memo = {}

    def f_at(a):                       # sub-generator: memoised chi^2(a)
        if a not in memo:
            memo[a] = yield a
        return memo[a]

    bracket = False
    alpha = alpha0 = 0.
    val = val0 = 1.
    sf_used = None
    nu = 0.
    for sf in SCALE_FACTORS:
        nu = npts * sf
        alpha0, val0, alpha = 0., 1., 0.
        val = (yield from f_at(alpha)) - nu
        if val < 0:
            return 'too_smooth', 0, dict(sf=sf)
        while val0 * val > 0:
            bracket = True
            val0 = val
            alpha0 = alpha
            alpha = alpha - 1.
            val = (yield from f_at(alpha)) - nu
            if alpha < -100.:
                bracket = False
                break
        if bracket:
            sf_used = sf
            break
    if not bracket:
        return 'no_root', float('nan'), dict(sf=None)
    found = None
    other_end = None
    if multisection:
        ms = multisection_gen(alpha, alpha0, val, val0, int(multisection))
        try:
            xs = next(ms)
            while True:
                chis = yield xs
                xs = ms.send([c - nu for c in chis])
        except StopIteration as stop:
            found = stop.value                  # None: not provably a single root -> the reference's iteration
    if found is not None:
        root, iters, _ = found
        finder = 'multisection'
    else:
        br = brentq_gen(alpha, alpha0, fa=val, fb=val0)
        try:
            x = next(br)
            while True:
                x = br.send((yield from f_at(x)) - nu)
        except StopIteration as stop:
            root, iters, _, other_end = stop.value
        finder = 'brentq'
    return 'root', float(np.power(10., root)), dict(sf=sf_used, bracket=(alpha, alpha0), log10_alpha=root,
                                                     iterations=iters, finder=finder, other_end=other_end)


def run_batched(npts_list, chi2_batch, prefetch=8, multisection=0):
    """Drive one search coroutine per record against a batched chi^2 evaluator.

    npts_list[i]: number of finite data points of record i (``len(b)``, interpolate.py:175), or None to
    skip the record (result NaN).  chi2_batch(rec_idx: int array, log10_alpha: float array) -> chi^2 array.
    Returns (alpha list, outcome list, info list, number of chi^2 evaluations).
    """
    T = len(npts_list)
    gens, pending = {}, {}
    cache = [dict() for _ in range(T)]
    results = [(None, float('nan'), {})] * T
    nevals = 0
    for i, n in enumerate(npts_list):
        if n is None:
            results[i] = ('skipped', float('nan'), {})
            continue
        g = chi2_search_gen(n, multisection=multisection)
        gens[i] = g
        pending[i] = next(g)

    def advance(i, value):
        try:
            pending[i] = gens[i].send(value)
        except StopIteration as stop:
            results[i] = stop.value
            del gens[i], pending[i]

    while gens:
        # serve everything already cached
        progressed = True
        while progressed:
            progressed = False
            for i in list(pending):
                a = pending[i]
                if isinstance(a, tuple):
                    if all(x in cache[i] for x in a):
                        advance(i, [cache[i][x] for x in a])
                        progressed = True
                elif a in cache[i]:
                    advance(i, cache[i][a])
                    progressed = True
        if not gens:
            break
        rec, alp = [], []
        for i, a in pending.items():
            if isinstance(a, tuple):
                for x in a:
                    if x not in cache[i]:
                        rec.append(i)
                        alp.append(x)
                continue
            rec.append(i)
            alp.append(a)
            # walk prefetch: integer alphas continue downwards; harmless extra evaluations
            if prefetch and a == math.floor(a) and -100. <= a <= 0.:
                for k in range(1, prefetch):
                    ak = a - k
                    if ak >= -101. and ak not in cache[i]:
                        rec.append(i)
                        alp.append(ak)
        vals = chi2_batch(np.asarray(rec, dtype=np.int32), np.asarray(alp, dtype=np.float64))
        nevals += len(rec)
        for i, a, v in zip(rec, alp, vals):
            cache[i][a] = float(v)
    return ([r[1] for r in results], [r[0] for r in results], [r[2] for r in results], nevals)


def run_brent_batched(brackets, f_batch):
    """Brent's iteration on given brackets, batched over records: brackets = {rec: (xa, xb, fa, fb)} with f values
    (chi^2 - nu) of opposite sign; f_batch(rec array, x array) -> f array.  Returns {rec: (root, iterations, other_end)}."""
    gens, pending, out = {}, {}, {}
    for i, (xa, xb, fa, fb) in brackets.items():
        g = brentq_gen(xa, xb, fa=fa, fb=fb)
        try:
            pending[i] = next(g)
            gens[i] = g
        except StopIteration as stop:
            out[i] = (stop.value[0], stop.value[1], stop.value[3])
    while gens:
        rec = np.array(sorted(gens), dtype=np.int32)
        xs = np.array([pending[int(i)] for i in rec], dtype=np.float64)
        vals = f_batch(rec, xs)
        for i, v in zip(rec.tolist(), vals):
            try:
                pending[i] = gens[i].send(float(v))
            except StopIteration as stop:
                out[i] = (stop.value[0], stop.value[1], stop.value[3])
                del gens[i], pending[i]
    return out
