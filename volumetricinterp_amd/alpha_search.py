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
import os

import numpy as np

SCALE_FACTORS = (0.6, 0.7, 0.8, 0.9, 1.0)      # interpolate.py:173
DECADES = tuple(float(-k) for k in range(102))  # log10(alpha) of the bracket walk, interpolate.py:186-203
XTOL = 2e-12
RTOL = 4 * np.finfo(np.float64).eps
MAXITER = 100

NO_ROOT_MSG = 'Could not find any roots to the objective function chi^2-nu in the range (1e-100,1).'


def _signbit(x):
    return math.copysign(1.0, x) < 0


JUMP_STOP_WIDTH = 1e-7      # decades of log10 alpha
JUMP_STOP_FRAC = 1e-4       # of nu (= FitEngine.REDO_TOL: beyond it the guard does not accept a record as a root anyway)


def jump_rule(nu):
    """The early end of Brent's iteration on a JUMP of chi^2(alpha) - nu (round 4).  Where an eigenvalue
    of X(alpha) crosses the truncation cut, chi^2 jumps: the function changes sign without a root, and brentq - in the
    reference too - bisects that sign change down to its xtol = 2e-12 in 40-60 function values (15 % of the synthetic
    default-order records; 44 ms against 16 ms for a record, and whichever workgroup draws one late ends the batch's launch
    alone).  The guard's polish already stops on such a record once the sign change is confined to 1e-7 decades
    (run_polish_batched): alpha means nothing beyond that - LAPACK drivers move the reference's own jumps by 1e-4 decades and
    more.  So the iteration ends when the bracket is narrower than JUMP_STOP_WIDTH while BOTH ends miss nu by more than
    JUMP_STOP_FRAC nu.  A genuine root never meets that: d chi^2 / d log10 alpha ~ nu per decade puts |f| ~ 1e-7 nu at that
    width.  One rule, four places: brentq_gen, BrentBatch._top, brent_top in csrc/vi_brent.hip (device kernel and the C loop).
    Returns (width, |f| threshold) or None (VINTERP_JUMP_STOP=0: brentq's own end, as in rounds 1-3)."""
    if os.environ.get('VINTERP_JUMP_STOP', '1') == '0':
        return None
    return (JUMP_STOP_WIDTH, JUMP_STOP_FRAC * float(nu))


def brentq_gen(xa, xb, fa=None, fb=None, xtol=XTOL, rtol=RTOL, maxiter=MAXITER, jump=None):
    """Coroutine form of brentq: yields x, receives f(x); returns (root, iterations, funcalls, other_end) with other_end
    the far end of the final bracket (None when an end point is an exact zero).

    fa / fb may be supplied when already known (the function is deterministic).  jump: None (brentq as SciPy has it) or
    (width, fmin) - see jump_rule."""
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
        if jump is not None and abs(xblk - xcur) <= jump[0] and abs(fcur) > jump[1]:      # |fblk| >= |fcur| here
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


class BrentBatch(object):
    """brentq_gen for many records at once: the same iteration, statement for statement, on arrays - one set of NumPy
    operations per round instead of one coroutine step per record (a batch of 1000 records spent a tenth of its time
    stepping 15 000 coroutine iterations).  Records join when their bracket is known (`add`), `requests()` lists the
    abscissae wanted, `feed()` takes the function values and advances; finished records are in `results`
    {rec: (root, iterations, funcalls, other_end)}.  Bit-identical to brentq_gen (tests/test_alpha_search.py)."""

    def __init__(self, n, xtol=XTOL, rtol=RTOL, maxiter=MAXITER):
        self.xtol, self.rtol, self.maxiter = xtol, rtol, maxiter
        z = lambda: np.zeros(n)
        self.xpre, self.xcur, self.xblk, self.fpre, self.fcur, self.fblk, self.spre, self.scur = (z() for _ in range(8))
        self.it = np.zeros(n, dtype=np.int64)
        self.funcalls = np.zeros(n, dtype=np.int64)
        self.active = np.zeros(n, dtype=bool)
        self.results = {}
        self.jump_width = 0.                        # jump_rule: 0 = off
        self.jump_fmin = np.full(n, np.inf)

    def add(self, i, xa, xb, fa, fb, jump=None):
        """brentq_gen(xa, xb, fa=fa, fb=fb, jump=jump) up to its first yield."""
        if jump is not None:
            self.jump_width = float(jump[0])
            self.jump_fmin[i] = float(jump[1])
        else:
            self.jump_fmin[i] = np.inf
        if fa == 0:
            self.results[i] = (xa, 0, 0, None)
            return
        if fb == 0:
            self.results[i] = (xb, 0, 0, None)
            return
        if _signbit(fa) == _signbit(fb):
            raise ValueError('f(a) and f(b) must have different signs')
        self.xpre[i], self.xcur[i], self.fpre[i], self.fcur[i] = xa, xb, fa, fb
        self.xblk[i] = self.fblk[i] = self.spre[i] = self.scur[i] = 0.
        self.it[i] = 1
        self.funcalls[i] = 0
        self.active[i] = True
        self._top(np.array([i]))

    def requests(self):
        idx = np.nonzero(self.active)[0]
        return idx, self.xcur[idx]

    def feed(self, idx, fvals):
        idx = np.asarray(idx, dtype=np.int64)
        self.fcur[idx] = fvals
        self.funcalls[idx] += 1
        self.it[idx] += 1
        if np.any(self.it[idx] > self.maxiter):
            raise RuntimeError('Failed to converge after %d iterations.' % self.maxiter)
        self._top(idx)

    def _top(self, idx):
        """One pass of brentq_gen's loop body for the records idx: from the top of the loop to the next yield."""
        xpre, xcur, xblk = self.xpre[idx], self.xcur[idx], self.xblk[idx]
        fpre, fcur, fblk = self.fpre[idx], self.fcur[idx], self.fblk[idx]
        spre, scur = self.spre[idx], self.scur[idx]
        c1 = (fpre != 0) & (fcur != 0) & (np.signbit(fpre) != np.signbit(fcur))
        xblk = np.where(c1, xpre, xblk)
        fblk = np.where(c1, fpre, fblk)
        d = xcur - xpre
        spre = np.where(c1, d, spre)
        scur = np.where(c1, d, scur)
        c2 = np.abs(fblk) < np.abs(fcur)
        xpre, xcur, xblk = np.where(c2, xcur, xpre), np.where(c2, xblk, xcur), np.where(c2, xcur, xblk)
        fpre, fcur, fblk = np.where(c2, fcur, fpre), np.where(c2, fblk, fcur), np.where(c2, fcur, fblk)
        delta = (self.xtol + self.rtol * np.abs(xcur)) / 2
        sbis = (xblk - xcur) / 2
        term = (fcur == 0) | (np.abs(sbis) < delta)
        if self.jump_width > 0.:
            term = term | ((np.abs(xblk - xcur) <= self.jump_width) & (np.abs(fcur) > self.jump_fmin[idx]))
        with np.errstate(all='ignore'):
            c3 = (np.abs(spre) > delta) & (np.abs(fcur) < np.abs(fpre))
            stry_sec = -fcur * (xcur - xpre) / (fcur - fpre)
            dpre = (fpre - fcur) / (xpre - xcur)
            dblk = (fblk - fcur) / (xblk - xcur)
            stry_iqi = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre))
            stry = np.where(xpre == xblk, stry_sec, stry_iqi)
            a, b = np.abs(spre), 3 * np.abs(sbis) - delta
            accept = c3 & (2 * np.abs(stry) < np.where(b < a, b, a))
        spre, scur = np.where(accept, scur, sbis), np.where(accept, stry, sbis)
        nxpre, nfpre = xcur, fcur
        nxcur = np.where(np.abs(scur) > delta, xcur + scur, xcur + np.where(sbis > 0, delta, -delta))
        # records that end here keep xcur (the root) and xblk (the other end); the others move on
        go = ~term
        self.xpre[idx] = np.where(go, nxpre, xpre)
        self.fpre[idx] = np.where(go, nfpre, fpre)
        self.xcur[idx] = np.where(go, nxcur, xcur)
        self.fcur[idx] = fcur
        self.xblk[idx], self.fblk[idx], self.spre[idx], self.scur[idx] = xblk, fblk, spre, scur
        for j in np.nonzero(term)[0].tolist():
            i = int(idx[j])
            self.active[i] = False
            self.results[i] = (float(xcur[j]), int(self.it[i]), int(self.funcalls[i]), float(xblk[j]))


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


class Exact(tuple):
    """A request for chi^2 at these log10(alpha) from the evaluator's reference-grade path (see chi2_search_gen)."""
    __slots__ = ()


# |chi^2 - nu| <= this fraction of nu at a walk point: the sign is not taken from an approximate walk value.  Measured
# (tools/exp_walk_floor.py, 1000 records x 49 decades of the BASELINE configs[2] geometry): the shared-basis chi^2 is within
# 3e-4 of the cold one on 99.9 % of the systems, median 4e-9, 53 of 50 000 beyond 3e-4, one beyond 1e-3 (1.01e-3), none
# beyond 3e-3 - next to the poles of chi^2(alpha), decades -28 and -33.  Five times the worst seen; it costs 1.7 extra cold
# solves per record (0.4 at 1e-3, 3.5 at 1e-2).
WALK_SIGN_MARGIN = 5e-3


def chi2_search_gen(npts, multisection=0, refine=False, prefetch=1, defer_brent=False):
    """Coroutine form of Interpolate.chi2 (interpolate.py:152-218) for one record.

    Yields log10(alpha) (or a tuple of them), receives chi^2 at that alpha (or a list).  Returns
    (outcome, alpha, info) with outcome in {'too_smooth', 'no_root', 'root'}; alpha is 0, NaN or 10**root as
    in the reference.  multisection = K > 0: try the guarded K-point multisection first (see multisection_gen),
    falling back to the Brent iteration when the bracket is not provably single-rooted.

    prefetch: a walk value that is not known yet is asked for together with the next prefetch - 1 decades below it (one
    tuple request; the extra evaluations are harmless, the reference would have reached most of them anyway).

    refine: the evaluator serves the bracket walk from a cheaper path whose chi^2 carries noise (the shared bases of
    FitEngine, ~1e-5 relative next to the poles of chi^2) and can serve ``Exact`` requests from its reference-grade one.
    The walk only decides SIGNS, so its values are used as they come unless one lies within WALK_SIGN_MARGIN of nu; the two
    ends of the bracket, whose values seed Brent's first steps, are always asked for again as Exact - the root finder
    then sees the same numbers as without the cheaper path.  refine = 'all': every walk value is an Exact request (the
    fallback when the refined ends contradict the walk)."""
    memo, memo_x = {}, {}

    def f_at(a):                       # sub-generator: memoised chi^2(a)
        if a not in memo:
            memo[a] = yield a
        return memo[a]

    all_exact = refine == 'all'
    margin = WALK_SIGN_MARGIN if (refine and not all_exact) else -1.
    if all_exact:
        # the whole table in one request (a walk of single requests would cost one evaluator round per decade)
        table = tuple(float(-k) for k in range(0, 102))
        for a, c in zip(table, (yield Exact(table))):
            memo_x[a] = c
    walk = memo_x if all_exact else memo

    def fetch(a):                      # a walk value that is not known yet, with the decades below it
        if prefetch <= 1:
            walk[a] = yield a
            return walk[a]
        ks = tuple(a - j for j in range(int(prefetch)) if a - j >= -101. and (a - j) not in walk)
        for k, v in zip(ks, (yield ks)):
            walk[k] = v
        return walk[a]

    cache_tab = [None, None]          # the trusted table and which of its entries are reference-grade

    def walk_sf(sf, nu):
        """The walk of one scale factor (interpolate.py:173-206): ('too_smooth' | 'bracket' | 'none', alpha, alpha0, val, val0).
        Values are the reference-grade ones where known, the evaluator's plain ones otherwise; a plain value within the sign
        margin of nu is not trusted: the walk is finished provisionally, ALL such decades are asked for as Exact in one
        request, and the walk is done again (asking one at a time cost an evaluator round each)."""
        while True:
            pend = []
            if len(walk) >= 102:
                # the whole table is known (the first scale factor usually walks to the end without a bracket): the same walk
                # in a few array operations instead of 102 interpreted steps
                if cache_tab[0] is None:
                    cache_tab[0] = np.array([memo_x[a] if a in memo_x else walk[a] for a in DECADES])
                    cache_tab[1] = np.array([a in memo_x for a in DECADES])
                tab, isx = cache_tab
                v = tab - nu
                entered = v[0] > 0                                  # val0 * val > 0 with val0 = 1
                stops = np.nonzero(~(v[:-1] * v[1:] > 0))[0]
                k = int(stops[0]) + 1 if len(stops) else 102        # first step whose product is not positive
                last = min(k, 101) if entered else 0
                if margin >= 0.:
                    if np.any(~(np.abs(v[:last + 1]) > margin * nu) & ~isx[:last + 1]):
                        # with the table complete, the doubtful decades of the later scale factors come along
                        doubt = np.zeros(102, dtype=bool)
                        for sf2 in SCALE_FACTORS:
                            if sf2 >= sf:
                                doubt |= ~(np.abs(tab - npts * sf2) > margin * npts * sf2)
                        pend = [DECADES[j] for j in np.nonzero(doubt & ~isx)[0].tolist()]
                if not pend:
                    if v[0] < 0:
                        return 'too_smooth', 0., 0., float(v[0]), 1.
                    if entered and k <= 100:                        # the step to -101 ends the walk without a bracket
                        return 'bracket', float(-k), float(-(k - 1)), float(v[k]), float(v[k - 1])
                    return 'none', 0., 0., 0., 0.
            else:
                # (the look-ups of the walk are written out: a sub-generator per step cost more than the step)
                kind = 'none'
                alpha0, val0, alpha = 0., 1., 0.
                c = memo_x.get(alpha)
                if c is None:
                    c = walk.get(alpha)
                    if c is None:
                        c = yield from fetch(alpha)
                    if margin >= 0. and not abs(c - nu) > margin * nu:
                        pend.append(alpha)
                val = c - nu
                if val < 0:
                    kind = 'too_smooth'
                else:
                    entered = False
                    while val0 * val > 0:
                        entered = True
                        val0 = val
                        alpha0 = alpha
                        alpha = alpha - 1.
                        c = memo_x.get(alpha)
                        if c is None:
                            c = walk.get(alpha)
                            if c is None:
                                c = yield from fetch(alpha)
                            if margin >= 0. and not abs(c - nu) > margin * nu:
                                pend.append(alpha)
                        val = c - nu
                        if alpha < -100.:
                            entered = False
                            break
                    if entered:
                        kind = 'bracket'
                if not pend:
                    return kind, alpha, alpha0, val, val0
            ks = tuple(pend)
            for a, c in zip(ks, (yield Exact(ks))):
                memo_x[a] = c
            cache_tab[0] = None

    bracket = False
    alpha = alpha0 = 0.
    val = val0 = 1.
    sf_used = None
    nu = 0.
    for sf in SCALE_FACTORS:
        nu = npts * sf
        kind, a_, a0_, v_, v0_ = yield from walk_sf(sf, nu)
        if kind == 'too_smooth':
            return 'too_smooth', 0, dict(sf=sf)
        if kind == 'bracket':
            bracket = True
            alpha, alpha0, val, val0 = a_, a0_, v_, v0_
            sf_used = sf
            break
    if not bracket:
        return 'no_root', float('nan'), dict(sf=None)
    if refine and refine != 'all':
        need = [a for a in (alpha, alpha0) if a not in memo_x]
        if need:
            for a, c in zip(need, (yield Exact(tuple(need)))):
                memo_x[a] = c
        va, vb = memo_x[alpha] - nu, memo_x[alpha0] - nu
        if not va * vb <= 0:
            # the reference-grade values do not bracket a sign change where the walk saw one: redo this record's walk on them
            out = yield from chi2_search_gen(npts, multisection=multisection, refine='all', defer_brent=defer_brent)
            out[2]['walk_redone_exact'] = True
            return out
        val, val0 = va, vb
        memo[alpha], memo[alpha0] = memo_x[alpha], memo_x[alpha0]       # Brent re-evaluates its ends through f_at
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
    elif defer_brent:
        # the driver runs Brent's iteration for all records at once (BrentBatch)
        return 'bracket', None, dict(sf=sf_used, alpha=alpha, alpha0=alpha0, val=val, val0=val0, nu=nu)
    else:
        br = brentq_gen(alpha, alpha0, fa=val, fb=val0, jump=jump_rule(nu))
        try:
            x = next(br)
            while True:
                x = br.send((yield from f_at(x)) - nu)
        except StopIteration as stop:
            root, iters, _, other_end = stop.value
        finder = 'brentq'
    return 'root', float(np.power(10., root)), dict(sf=sf_used, bracket=(alpha, alpha0), log10_alpha=root,
                                                     iterations=iters, finder=finder, other_end=other_end)


def run_batched(npts_list, chi2_batch, prefetch=8, multisection=0, refine=False, vector_brent=True, brent_solver=None):
    """Drive one search coroutine per record against a batched chi^2 evaluator.

    npts_list[i]: number of finite data points of record i (``len(b)``, interpolate.py:175), or None to
    skip the record (result NaN).  chi2_batch(rec_idx: int array, log10_alpha: float array) -> chi^2 array; with
    refine (see chi2_search_gen) it is also called as chi2_batch(rec, log10_alpha, exact: bool array).
    vector_brent: Brent's iteration of all records on arrays (BrentBatch) instead of one coroutine step per record and
    round; same iterates, same answers.
    brent_solver(records, brackets) -> list of (root, iterations, funcalls, other_end) or None per record: Brent's whole
    iteration for many records at once somewhere else (FitEngine: one kernel launch, a workgroup per record).  It is
    called once, when every record's walk has ended; a record it answers None for is iterated here.
    Returns (alpha list, outcome list, info list, number of chi^2 evaluations).
    """
    T = len(npts_list)
    brent = BrentBatch(T) if (vector_brent or brent_solver is not None) else None
    brackets = {}
    deferred = {}
    nu_arr = np.zeros(T)
    gens, pending = {}, {}
    cache = [dict() for _ in range(T)]
    cache_x = [dict() for _ in range(T)]
    results = [(None, float('nan'), {})] * T
    nevals = 0
    for i, n in enumerate(npts_list):
        if n is None:
            results[i] = ('skipped', float('nan'), {})
            continue
        g = chi2_search_gen(n, multisection=multisection, refine=refine, prefetch=prefetch, defer_brent=brent is not None)
        gens[i] = g
        pending[i] = next(g)

    def finish_brent(i):
        root, iters, _, other_end = brent.results.pop(i)
        b = brackets.pop(i)
        info = dict(sf=b['sf'], bracket=(b['alpha'], b['alpha0']), log10_alpha=root, iterations=iters, finder='brentq',
                    other_end=other_end)
        if b.get('walk_redone_exact'):
            info['walk_redone_exact'] = True
        results[i] = ('root', float(np.power(10., root)), info)

    def advance(i, value):
        try:
            pending[i] = gens[i].send(value)
        except StopIteration as stop:
            del gens[i], pending[i]
            if stop.value[0] == 'bracket':          # the walk is done: Brent's iteration goes on in the batch
                b = brackets[i] = stop.value[2]
                nu_arr[i] = b['nu']
                if brent_solver is not None and b['val'] != 0 and b['val0'] != 0:
                    deferred[i] = b                 # iterated with all the others once the walks are over
                    return
                brent.add(i, b['alpha'], b['alpha0'], b['val'], b['val0'], jump=jump_rule(b['nu']))
                if i in brent.results:
                    finish_brent(i)
            else:
                results[i] = stop.value

    def serve(i):
        # answer record i's requests from what is already known, for as long as that is possible
        while i in gens:
            a = pending[i]
            if isinstance(a, tuple):
                ci = cache_x[i] if isinstance(a, Exact) else cache[i]
                if all(x in ci for x in a):
                    advance(i, [ci[x] for x in a])
                else:
                    return
            elif a in cache[i]:
                advance(i, cache[i][a])
            else:
                return

    def serve_brent():
        # Brent requests whose value is already known (chi^2 is memoised per record, as in the coroutine)
        while True:
            idx, xs = brent.requests()
            hit = [(int(i), x) for i, x in zip(idx.tolist(), xs.tolist()) if x in cache[int(i)]]
            if not hit:
                return idx, xs
            ii = np.array([i for i, _ in hit])
            brent.feed(ii, np.array([cache[i][x] - brackets[i]['nu'] for i, x in hit]))
            for i, _ in hit:
                if i in brent.results:
                    finish_brent(i)

    for i in list(gens):
        serve(i)
    while gens or deferred or (brent is not None and brent.active.any()):
        if deferred and not gens:
            ids = sorted(deferred)
            for i, r_ in zip(ids, brent_solver(ids, [deferred[i] for i in ids])):
                b = deferred.pop(i)
                if r_ is None:
                    brent.add(i, b['alpha'], b['alpha0'], b['val'], b['val0'], jump=jump_rule(b['nu']))
                else:
                    brent.results[i] = r_
                    nevals += int(r_[2])
                if i in brent.results:
                    finish_brent(i)
            continue
        rec, alp, exact = [], [], []
        bidx = None
        if brent is not None and brent.active.any():
            bidx, bxs = serve_brent()
            rec += bidx.tolist()
            alp += bxs.tolist()
            exact += [False] * len(bidx)
            if not rec and not gens:
                break
        for i, a in pending.items():
            if isinstance(a, tuple):
                ex = isinstance(a, Exact)
                ci = cache_x[i] if ex else cache[i]
                for x in a:
                    if x not in ci:
                        rec.append(i)
                        alp.append(x)
                        exact.append(ex)
                continue
            exact.append(False)
            rec.append(i)
            alp.append(a)
        if any(exact):
            vals = chi2_batch(np.asarray(rec, dtype=np.int32), np.asarray(alp, dtype=np.float64),
                              np.asarray(exact, dtype=bool))
        else:
            vals = chi2_batch(np.asarray(rec, dtype=np.int32), np.asarray(alp, dtype=np.float64))
        nevals += len(rec)
        for i, a, v, ex in zip(rec, alp, np.asarray(vals, dtype=np.float64).tolist(), exact):
            (cache_x[i] if ex else cache[i])[a] = v
        if bidx is not None and len(bidx):
            nb = len(bidx)
            fv = np.asarray(vals[:nb], dtype=np.float64) - nu_arr[bidx]
            brent.feed(bidx, fv)
            for i in bidx.tolist():
                if i in brent.results:
                    finish_brent(i)
        for i in list(pending):             # every pending record had a request in this batch
            serve(i)
    return ([r[1] for r in results], [r[0] for r in results], [r[2] for r in results], nevals)


def run_table_batched(npts_list, chi2_batch, prefetch=8, refine=False, brent_solver=None):
    """run_batched for large batches: the same search, record by record the same requests and the same decisions, but the
    bracket walk of ALL records is carried by array operations on one (records x 102 decades) table instead of one coroutine
    per record - a batch of 1000 records spent 110 ms of 460 (and, the interpreter being one, of every concurrent pipeline)
    stepping 6 000 coroutine resumptions, 100 000 dictionary look-ups and 50 000 small NumPy calls through the walks.

    It restates chi2_search_gen (multisection = 0, Brent deferred) as a state machine over arrays:
      * a record walks its current scale factor on what is known of its table; if the walk runs off the known part the next
        `prefetch` decades are requested (chi2_search_gen.fetch);
      * with refine, walk values within WALK_SIGN_MARGIN of nu are asked for again as exact ones before the walk's outcome is
        used - the visited doubtful decades when the walk started on an incomplete table, the doubtful decades of this and
        all later scale factors when it started on a complete one (the two branches of chi2_search_gen.walk_sf);
      * the bracket ends are asked for as exact values; if they do not bracket a sign change the record's search is redone on
        a table of exact values only (refine = 'all');
      * Brent's iteration as in run_batched: brent_solver for all records at once, BrentBatch for what it leaves.
    tests/test_alpha_search.py holds it to run_batched on noisy synthetic tables, request for request."""
    T = len(npts_list)
    NSF = len(SCALE_FACTORS)
    SF = np.array(SCALE_FACTORS)
    dec = np.array(DECADES)
    cols = np.arange(102)
    live = np.array([n is not None for n in npts_list], dtype=bool)
    npts = np.array([0. if n is None else float(n) for n in npts_list])
    tab = np.full((T, 102), np.nan)          # plain values (chi2_search_gen's memo) and which of them are known
    kn = np.zeros((T, 102), dtype=bool)
    tabx = np.full((T, 102), np.nan)         # exact values (memo_x)
    isx = np.zeros((T, 102), dtype=bool)
    sfi = np.zeros(T, dtype=np.int64)        # scale factor being walked
    entry_tab = np.zeros(T, dtype=bool)      # this walk_sf pass started on a complete table
    fresh = live.copy()                      # a walk_sf pass starts: entry_tab is to be taken
    mode_all = np.zeros(T, dtype=bool)       # refine = 'all': the search redone on exact values
    walking = live.copy()
    ends = np.zeros(T, dtype=bool)           # bracket found, exact ends asked for
    brk_k = np.zeros(T, dtype=np.int64)
    results = [('skipped', float('nan'), {}) if n is None else (None, float('nan'), {}) for n in npts_list]
    brackets = {}
    nevals = 0
    margin0 = WALK_SIGN_MARGIN if refine else -1.

    def request(mask_plain, mask_exact):
        nonlocal nevals
        rp, cp = np.nonzero(mask_plain)
        rx, cx = np.nonzero(mask_exact)
        if len(rp) + len(rx) == 0:
            return
        rec = np.concatenate([rp, rx]).astype(np.int32)
        alp = np.concatenate([dec[cp], dec[cx]])
        if len(rx):
            ex = np.concatenate([np.zeros(len(rp), dtype=bool), np.ones(len(rx), dtype=bool)])
            vals = np.asarray(chi2_batch(rec, alp, ex), dtype=np.float64)
        else:
            vals = np.asarray(chi2_batch(rec, alp), dtype=np.float64)
        nevals += len(rec)
        tab[rp, cp] = vals[:len(rp)]
        kn[rp, cp] = True
        tabx[rx, cx] = vals[len(rp):]
        isx[rx, cx] = True

    def bracket_found(i, k, val, val0, sf, nu):
        b = dict(sf=sf, alpha=float(-k), alpha0=float(-(k - 1)), val=float(val), val0=float(val0), nu=nu)
        if mode_all[i]:
            b['walk_redone_exact'] = True
        brackets[i] = b

    def finish(i, res):
        if mode_all[i]:
            res[2]['walk_redone_exact'] = True
        results[i] = res

    while walking.any() or ends.any():
        want_plain = np.zeros((T, 102), dtype=bool)
        want_exact = np.zeros((T, 102), dtype=bool)
        # ---- records whose exact bracket ends have arrived
        for i in np.nonzero(ends)[0].tolist():
            k = int(brk_k[i])
            sf = SCALE_FACTORS[sfi[i]]
            nu = npts_list[i] * sf
            va, vb = tabx[i, k] - nu, tabx[i, k - 1] - nu
            ends[i] = False
            if not va * vb <= 0:
                # the reference-grade values do not bracket a sign change where the walk saw one: the walk again, on them
                mode_all[i] = True
                sfi[i] = 0
                walking[i] = True
                fresh[i] = True
                want_exact[i] = ~isx[i]
            else:
                bracket_found(i, k, va, vb, sf, nu)
        if want_exact.any():
            request(want_plain, want_exact)
            want_exact[:] = False
        # ---- the walks, until every walking record has asked for something or is through
        todo = walking.copy()
        while todo.any():
            W = np.nonzero(todo)[0]
            full = kn[W].all(axis=1) | mode_all[W]
            entry_tab[W] = np.where(fresh[W], full, entry_tab[W])
            fresh[W] = False
            alle = mode_all[W]
            m = np.where(isx[W] | alle[:, None], tabx[W], tab[W])
            known = kn[W] | isx[W]
            margin = np.where(alle, -1., margin0)
            nu = npts[W] * SF[sfi[W]]
            v = m - nu[:, None]
            unk = ~known
            u = np.where(unk.any(axis=1), np.argmax(unk, axis=1), 102)          # first decade not known
            with np.errstate(invalid='ignore'):
                entered = v[:, 0] > 0
                bad = ~(v[:, :-1] * v[:, 1:] > 0)                                 # step j -> j + 1 ends the walk
            bad &= (cols[None, 1:] < u[:, None])
            hasstop = bad.any(axis=1)
            k = np.where(hasstop, np.argmax(bad, axis=1) + 1, 102)
            need0 = u == 0
            need_more = ~need0 & entered & ~hasstop & (u < 102)
            last = np.where(entered, np.minimum(k, 101), 0)
            ok = ~need0 & ~need_more
            # doubtful walk values (refine)
            with np.errstate(invalid='ignore'):
                dsel = (cols[None, :] <= last[:, None]) & ~isx[W] & ~(np.abs(v) > (margin * nu)[:, None]) & (margin >= 0.)[:, None]
            dsel &= ok[:, None]
            trig = dsel.any(axis=1)
            if trig.any():
                pend = dsel.copy()
                tt = trig & entry_tab[W]
                if tt.any():
                    # with the table complete, the doubtful decades of the later scale factors come along
                    doubt = np.zeros((int(tt.sum()), 102), dtype=bool)
                    Wt = W[tt]
                    mt = m[tt]
                    for s2 in range(NSF):
                        sel = sfi[Wt] <= s2
                        with np.errstate(invalid='ignore'):
                            d2 = ~(np.abs(mt - (npts[Wt] * SCALE_FACTORS[s2])[:, None])
                                   > (margin0 * npts[Wt] * SCALE_FACTORS[s2])[:, None])
                        doubt |= d2 & sel[:, None]
                    pend[tt] = doubt & ~isx[Wt]
                want_exact[W[trig]] = pend[trig]
                fresh[W[trig]] = True                    # the pass starts again when the exact values are in
                todo[W[trig]] = False
            # records that need more of their table
            for sel, first in ((need0, None), (need_more, u)):
                if sel.any():
                    Ws = W[sel]
                    a0 = np.zeros(len(Ws), dtype=np.int64) if first is None else first[sel]
                    chunk = (cols[None, :] >= a0[:, None]) & (cols[None, :] < a0[:, None] + int(max(1, prefetch))) & ~kn[Ws]
                    want_plain[Ws] = chunk
                    todo[Ws] = False
            # walks that are through
            done = ok & ~trig
            if done.any():
                idx = np.nonzero(done)[0]
                for j in idx.tolist():
                    i = int(W[j])
                    sf = SCALE_FACTORS[sfi[i]]
                    if v[j, 0] < 0:
                        finish(i, ('too_smooth', 0, dict(sf=sf)))
                        walking[i] = todo[i] = False
                    elif entered[j] and k[j] <= 100:
                        kk = int(k[j])
                        walking[i] = todo[i] = False
                        if refine and not mode_all[i]:
                            brk_k[i] = kk
                            ends[i] = True
                            want_exact[i, kk] = not isx[i, kk]
                            want_exact[i, kk - 1] = not isx[i, kk - 1]
                        else:
                            bracket_found(i, kk, v[j, kk], v[j, kk - 1], sf, npts_list[i] * sf)
                    else:
                        sfi[i] += 1
                        fresh[i] = True
                        if sfi[i] >= NSF:
                            finish(i, ('no_root', float('nan'), dict(sf=None)))
                            walking[i] = todo[i] = False
        request(want_plain, want_exact)

    # ---- Brent's iteration (as in run_batched)
    brent = BrentBatch(T)
    nu_arr = np.zeros(T)
    deferred = {}
    bcache = {}

    def finish_brent(i):
        root, iters, _, other_end = brent.results.pop(i)
        b = brackets.pop(i)
        info = dict(sf=b['sf'], bracket=(b['alpha'], b['alpha0']), log10_alpha=root, iterations=iters, finder='brentq',
                    other_end=other_end)
        if b.get('walk_redone_exact'):
            info['walk_redone_exact'] = True
        results[i] = ('root', float(np.power(10., root)), info)

    for i in sorted(brackets):
        b = brackets[i]
        nu_arr[i] = b['nu']
        if brent_solver is not None and b['val'] != 0 and b['val0'] != 0:
            deferred[i] = b
            continue
        brent.add(i, b['alpha'], b['alpha0'], b['val'], b['val0'], jump=jump_rule(b['nu']))
        if i in brent.results:
            finish_brent(i)
    if deferred:
        ids = sorted(deferred)
        for i, r_ in zip(ids, brent_solver(ids, [deferred[i] for i in ids])):
            b = deferred.pop(i)
            if r_ is None:
                brent.add(i, b['alpha'], b['alpha0'], b['val'], b['val0'], jump=jump_rule(b['nu']))
            else:
                brent.results[i] = r_
                nevals += int(r_[2])
            if i in brent.results:
                finish_brent(i)
    while brent.active.any():
        idx, xs = brent.requests()
        hit = [(int(i), x) for i, x in zip(idx.tolist(), xs.tolist()) if x in bcache.get(int(i), ())]
        if hit:
            brent.feed(np.array([i for i, _ in hit]), np.array([bcache[i][x] - nu_arr[i] for i, x in hit]))
            for i, _ in hit:
                if i in brent.results:
                    finish_brent(i)
            continue
        vals = np.asarray(chi2_batch(idx.astype(np.int32), np.asarray(xs, dtype=np.float64)), dtype=np.float64)
        nevals += len(idx)
        for i, x, c in zip(idx.tolist(), xs.tolist(), vals.tolist()):
            bcache.setdefault(i, {})[x] = c
        brent.feed(idx, vals - nu_arr[idx])
        for i in idx.tolist():
            if i in brent.results:
                finish_brent(i)
    return ([r[1] for r in results], [r[0] for r in results], [r[2] for r in results], nevals)


POLISH_XTOL = 1e-7            # decades: a sign change confined to less than this without |f| getting small is a jump
POLISH_BRENT_ROUNDS = 6


def run_polish_batched(brackets, f_batch, ftol, target=None, xtol=POLISH_XTOL, brent_rounds=POLISH_BRENT_ROUNDS,
                       ksection=15, skip_brent=()):
    """Root polishing of the consistency guard (FitEngine._search_and_finalize): small brackets around an approximate
    root, an expensive f (cold solves), few records.  brackets = {rec: (xa, xb, fa, fb)}, f_batch(rec, x) -> f,
    ftol = {rec: |f| below which the record is done}, target = {rec: x} the point whose nearest sign change is wanted.

    Unlike the search proper this is not a restatement of the reference's brentq call: alpha is only meaningful to ~1e-6
    decades here (the noise of chi^2 next to the poles and jumps the guard deals with), so a record stops as soon as
    |f| <= ftol - a root for the guard's purposes - or its sign change is confined to xtol decades without |f| ever
    getting there: a jump of f (an eigenvalue of X(alpha) crossing the truncation threshold), where bisecting on to
    brentq's 2e-12 would cost 20 more rounds of one cold solve each for nothing.  Rounds are what costs (every one is a
    launch that lasts as long as one cold solve): Brent's steps first (superlinear on smooth functions), at most
    `brent_rounds` of them, then K-section with `ksection` points per record and round (a fixed number: what a record gets
    must not depend on how many others are being polished alongside).  skip_brent: records that go to the K-section at once -
    the guard lists those whose bracket shows the signature of a jump (both ends far from zero on a short bracket), where
    Brent can only bisect: 16-fold per round instead of 2-fold (round 4: the jump record of the bench spent 8 rounds here).
    A record's sequence of abscissae depends on its own bracket only; records in the two phases share the rounds' launches.
    Returns {rec: (root, rounds, other_end, how)}, how in {'ftol', 'jump', 'xtol'}."""
    out = {}
    state = {}                              # rec -> [lo, hi, flo, fhi, best_x, best_f]
    gens, pending = {}, {}
    ksec = []
    rounds = 0

    def note(i, x, f):
        st = state[i]
        if math.isnan(f):
            return
        if abs(f) < abs(st[5]):
            st[4], st[5] = x, f
        if st[0] < x < st[1]:
            if _signbit(f) == _signbit(st[2]):
                st[0], st[2] = x, f
            else:
                st[1], st[3] = x, f

    def finished(i):
        st = state[i]
        if abs(st[5]) <= ftol[i]:
            out[i] = (st[4], rounds, st[1] if st[4] == st[0] else st[0], 'ftol')
        elif st[1] - st[0] <= xtol:
            lo_best = abs(st[2]) <= abs(st[3])
            out[i] = (st[0] if lo_best else st[1], rounds, st[1] if lo_best else st[0], 'jump')
        else:
            return False
        return True

    for i, (xa, xb, fa, fb) in brackets.items():
        lo, hi, flo, fhi = (xa, xb, fa, fb) if xa < xb else (xb, xa, fb, fa)
        state[i] = [lo, hi, flo, fhi, lo if abs(flo) <= abs(fhi) else hi, min(abs(flo), abs(fhi))]
        if finished(i):
            continue
        if i in skip_brent:
            ksec.append(i)
            continue
        g = brentq_gen(xa, xb, fa=fa, fb=fb)
        try:
            pending[i] = next(g)
            gens[i] = g
        except StopIteration as stop:
            out[i] = (stop.value[0], 0, stop.value[3], 'xtol')
    K = int(ksection)
    while gens or ksec:
        if gens and rounds >= brent_rounds:             # Brent's share is over: the records still iterating go to the K-section
            ksec = sorted(set(ksec) | set(gens))
            gens.clear()
            pending.clear()
        brec = sorted(gens)
        rec = list(brec)
        xs = [pending[i] for i in brec]
        for i in ksec:
            lo, hi = state[i][0], state[i][1]
            for k in range(K):
                rec.append(i)
                xs.append(lo + (hi - lo) * (k + 1) / (K + 1))
        vals = f_batch(np.asarray(rec, dtype=np.int32), np.asarray(xs, dtype=np.float64))
        rounds += 1
        for i, x, v in zip(brec, xs[:len(brec)], vals[:len(brec)]):
            note(i, x, float(v))
            if finished(i):
                del gens[i], pending[i]
                continue
            try:
                pending[i] = gens[i].send(float(v))
            except StopIteration as stop:
                out[i] = (stop.value[0], rounds, stop.value[3], 'xtol')
                del gens[i], pending[i]
        o = len(brec)
        for n, i in enumerate(ksec):
            st = state[i]
            pts = ([(st[0], st[2])] + [(xs[o + n * K + k], float(vals[o + n * K + k])) for k in range(K)] + [(st[1], st[3])])
            pts = [p for p in pts if not math.isnan(p[1])]
            for x, f in pts:
                if abs(f) < abs(st[5]):
                    st[4], st[5] = x, f
            tgt = target[i] if target is not None and i in target else st[4]
            best = None
            for (xa, fa), (xb, fb) in zip(pts[:-1], pts[1:]):
                if _signbit(fa) != _signbit(fb):
                    d = min(abs(xa - tgt), abs(xb - tgt)) if not (xa <= tgt <= xb) else 0.
                    if best is None or d < best[0]:
                        best = (d, xa, xb, fa, fb)
            if best is not None:
                st[0], st[1], st[2], st[3] = best[1:]
            else:                               # cannot happen with finite end values of opposite sign; do not loop
                st[1] = st[0]
        ksec = [i for i in ksec if not finished(i)]
    return out
