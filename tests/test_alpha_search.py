"""Host-side scalar logic of the regularisation-parameter search (gate L5) and the brentq restatement."""
import math

import numpy as np
import pytest
import scipy.optimize

from conftest import load_golden
from volumetricinterp_amd import alpha_search as AS


def drive(gen_fn, f):
    """Run a coroutine against a plain function; returns (result, list of x requested)."""
    g = gen_fn
    xs = []
    try:
        x = next(g)
        while True:
            xs.append(x)
            x = g.send(f(x))
    except StopIteration as stop:
        return stop.value, xs


FUNCS = [
    (lambda x: x**3 - 2 * x - 5, 2., 3.),
    (lambda x: math.cos(x) - x, 0., 1.),
    (lambda x: math.exp(-x) - 1e-3 * x**2 - 0.5, -1., 4.),
    (lambda x: (x - 0.3) * (1 + 50 * (x - 0.3)**2), -1., 1.),          # odd, steep
    (lambda x: math.tanh(40 * (x + 20.37)) * 500 + 3, -21., -20.),     # step-like, as chi2-nu can be
    (lambda x: 1e-9 * (x + 30.37), -31., -30.),
    (lambda x: x, -1., 0.),                                             # root at the bracket end
]


@pytest.mark.parametrize('case', range(len(FUNCS)))
def test_brentq_restatement_matches_scipy_call_for_call(case):
    f, a, b = FUNCS[case]
    calls = []

    def logged(x):
        calls.append(x)
        return f(x)
    ref, info = scipy.optimize.brentq(logged, a, b, full_output=True, disp=True)
    (root, iters, funcalls, _oe), xs = drive(AS.brentq_gen(a, b), f)
    assert root == ref                      # bit-identical iterate sequence
    assert xs == calls
    assert funcalls == info.function_calls
    if info.function_calls > 2:             # SciPy leaves `iterations` uninitialised when an end point is the root
        assert iters == info.iterations
    # supplying the known end values skips exactly the two initial evaluations
    (root2, _, _, _), xs2 = drive(AS.brentq_gen(a, b, fa=f(a), fb=f(b)), f)
    assert root2 == ref and xs2 == calls[2:]


def test_brentq_sign_error():
    with pytest.raises(ValueError, match='different signs'):
        drive(AS.brentq_gen(0., 1.), lambda x: 1 + x)


def _split_calls(calls, npts):
    """Split the reference's chi2objfunct log (alpha, nu, value) into per-record lookup tables."""
    recs, cur, t = [], None, 0
    for a, nu, v in calls:
        if a == 0.0 and t < len(npts) and abs(nu - 0.6 * npts[t]) < 1e-9 and (cur is None or len(cur) > 0):
            cur = {}
            recs.append(cur)
            t += 1
        cur[a] = v + nu                      # chi^2 itself
    return recs


@pytest.mark.parametrize('name', ['fit_k8l2', 'fit_k8l2_c2', 'fit_k8l2_psi', 'fit_default', 'fit_edge'])
def test_search_logic_on_reference_chi2_values(name):
    """Gate L5: fed the reference's own chi^2 values, the coroutine takes the same scale factor, the same
    bracket, requests only alphas the reference evaluated, and lands on the same root (|dlog10| <= 1e-9)."""
    f = load_golden(name)
    npts = [int(np.isfinite(v).sum()) for v in f['value']]
    tables = _split_calls(f['chi2_calls'], npts)
    assert len(tables) == len(npts)
    for t, (n, tab) in enumerate(zip(npts, tables)):
        missing = []

        def chi2(a):
            if a not in tab:
                missing.append(a)
                return float('nan')
            return tab[a]
        (outcome, alpha, info), xs = drive(AS.chi2_search_gen(n), chi2)
        assert not missing, (t, missing[:3])
        aref = f['alpha'][t]
        if np.isnan(aref):
            assert outcome == 'no_root' and np.isnan(alpha)
            assert len(set(xs)) == 102                       # alpha = 0 .. -101 (memoised across scale factors)
        elif aref == 0:
            assert outcome == 'too_smooth' and alpha == 0
        else:
            assert outcome == 'root'
            assert abs(math.log10(alpha) - math.log10(aref)) <= 1e-9
        # every distinct alpha the reference evaluated for this record is requested exactly once
        assert sorted(set(xs)) == sorted(tab.keys()) and len(xs) == len(set(xs))


def test_run_batched_matches_sequential():
    rng = np.random.default_rng(0)
    T = 7
    roots = rng.uniform(-40, -5, T)
    npts = [550] * T
    npts[3] = None                                            # skipped record

    def chi2_of(i, a):
        # smooth, decreasing in -alpha ... crosses 0.6 * 550 at roots[i]
        return 330. + 2000. * math.tanh(0.3 * (a - roots[i]))

    counts = []

    def batch(rec, alp):
        counts.append(len(rec))
        return np.array([chi2_of(i, a) for i, a in zip(rec, alp)])
    alphas, outcomes, infos, nev = AS.run_batched(npts, batch, prefetch=8)
    for i in range(T):
        if npts[i] is None:
            assert outcomes[i] == 'skipped' and np.isnan(alphas[i])
            continue
        (o, a, info), _ = drive(AS.chi2_search_gen(npts[i]), lambda x: chi2_of(i, x))
        assert outcomes[i] == o == 'root' and alphas[i] == a
        assert abs(math.log10(a) - roots[i]) < 1e-10
    assert nev == sum(counts)
    # prefetching must cut the number of GPU round trips well below the sequential walk length
    assert len(counts) < 25


def drive_ms(g, f):
    """Run a multisection coroutine; returns (result or None, number of rounds, K seen)."""
    rounds = 0
    try:
        xs = next(g)
        while True:
            rounds += 1
            xs = g.send([f(x) for x in xs])
    except StopIteration as stop:
        return stop.value, rounds


@pytest.mark.parametrize('case', range(len(FUNCS) - 1))
@pytest.mark.parametrize('K', [15, 255])
def test_multisection_finds_brents_root(case, K):
    f, a, b = FUNCS[case]
    ref = scipy.optimize.brentq(f, a, b)
    res, rounds = drive_ms(AS.multisection_gen(a, b, f(a), f(b), K), f)
    root, nr, calls = res
    # stops at a 1e-7 bracket and finishes with the secant point: far inside the 1e-7 parity tolerance on log10 alpha
    assert abs(root - ref) <= 1e-10
    assert nr == rounds and calls == K * rounds
    assert rounds <= math.ceil(math.log((b - a) / AS.MS_XTOL) / math.log(K + 1))


def test_multisection_guard_hands_multi_root_brackets_back():
    # three roots inside the bracket (measured shape of chi^2 - nu on the screened MAXK=8, MAXL=2 fixture): which one
    # brentq returns depends on its iterates, so multisection must refuse and the search must equal plain Brent
    f = lambda x: (x + 28.4698) * (x + 28.0745) * (x + 28.2)                  # noqa: E731
    res, rounds = drive_ms(AS.multisection_gen(-29., -28., f(-29.), f(-28.), 255), f)
    assert res is None and rounds == 1
    # NaN or an exact zero among the samples also hands back
    res, _ = drive_ms(AS.multisection_gen(-1., 1., -1., 1., 3), lambda x: x)       # sample at exactly 0
    assert res is None
    res, _ = drive_ms(AS.multisection_gen(-1., 1., -1., 1., 4), lambda x: float('nan') if x > 0.5 else x - 0.1)
    assert res is None
    # end points that are already roots / same sign: Brent's business
    assert drive_ms(AS.multisection_gen(-1., 0., -1., 0., 15), lambda x: x)[0] is None
    assert drive_ms(AS.multisection_gen(-1., 0., 1., 2., 15), lambda x: x)[0] is None


def test_search_falls_back_to_brent_on_multi_root_bracket():
    # chi^2 - nu with three crossings in [-29, -28]; nu = 0.6 * 550
    chi2 = lambda a: 330. + 4e3 * (a + 28.4698) * (a + 28.0745) * (a + 28.2) if a < -27.5 else 330. + 50. * (a + 28.6)  # noqa: E731
    (o1, a1, i1), _ = drive(AS.chi2_search_gen(550), chi2)
    g = AS.chi2_search_gen(550, multisection=255)
    try:
        x = next(g)
        while True:
            x = g.send([chi2(v) for v in x] if isinstance(x, tuple) else chi2(x))
    except StopIteration as stop:
        o2, a2, i2 = stop.value
    assert o1 == o2 == 'root' and a1 == a2                   # bit-identical: same Brent iterates
    assert i1['finder'] == i2['finder'] == 'brentq'


def test_search_with_multisection_on_smooth_chi2():
    roots = [-12.3456789, -33.3]
    for rt in roots:
        chi2 = lambda a: 330. + 2000. * math.tanh(0.3 * (a - rt))       # noqa: E731
        (o1, a1, _), xs1 = drive(AS.chi2_search_gen(550), chi2)
        g = AS.chi2_search_gen(550, multisection=63)
        nreq = 0
        try:
            x = next(g)
            while True:
                nreq += 1 if isinstance(x, tuple) else 0
                x = g.send([chi2(v) for v in x] if isinstance(x, tuple) else chi2(x))
        except StopIteration as stop:
            o2, a2, info = stop.value
        assert o1 == o2 == 'root'
        assert abs(math.log10(a1) - math.log10(a2)) <= 1e-10
        assert info['finder'] == 'multisection'
        assert nreq <= 4                            # 64^-4 < 1e-7: dependent rounds after the walk


def _literal_walk(chi2, npts):
    """interpolate.py:173-206 written out step by step (the bracket walk only): (outcome, sf, alpha, alpha0)."""
    bracket = False
    for sf in AS.SCALE_FACTORS:
        nu = npts * sf
        alpha0, val0, alpha = 0., 1., 0.
        val = chi2(alpha) - nu
        if val < 0:
            return 'too_smooth', sf, None, None
        while val0 * val > 0:
            bracket = True
            val0, alpha0 = val, alpha
            alpha = alpha - 1.
            val = chi2(alpha) - nu
            if alpha < -100.:
                bracket = False
                break
        if bracket:
            return 'root', sf, alpha, alpha0
    return 'no_root', None, None, None


@pytest.mark.parametrize('refine', [False, True])
def test_walk_over_a_known_table_equals_the_step_by_step_walk(refine):
    """Once a record's 102 walk values are known the coroutine walks the remaining scale factors with array operations;
    on random tables - sign changes anywhere (also in the last step, which the reference ignores), exact zeros, NaNs,
    values within the sign margin of nu - it must take the scale factor and the bracket of the literal loop.  With
    `refine` the evaluator returns the same numbers for Exact requests, so the answers must not change either."""
    rng = np.random.default_rng(7)
    npts = 1000
    for case in range(400):
        lvl = rng.choice([500., 650., 850., 950., 1050.])
        tab = lvl + rng.choice([1., 30., 300.]) * rng.standard_normal(102)
        if case % 3 == 0:                                     # a table that crosses the scale factors' targets somewhere
            k = rng.integers(1, 102)
            tab[k:] -= rng.uniform(100., 600.)
        if case % 5 == 0:
            tab[rng.integers(0, 102)] = float('nan')
        if case % 7 == 0:
            tab[rng.integers(0, 102)] = npts * rng.choice(AS.SCALE_FACTORS)          # an exact zero of chi^2 - nu
        if case % 11 == 0:
            tab[rng.integers(0, 102)] = npts * rng.choice(AS.SCALE_FACTORS) * (1. + 3e-4)     # inside the sign margin
        if case % 13 == 0:
            tab[0] = rng.uniform(100., 590.)                  # too smooth at once
        chi2 = lambda a: float(tab[int(round(-a))])
        want = _literal_walk(chi2, npts)
        g = AS.chi2_search_gen(npts, refine=refine)
        got = None
        try:
            x = next(g)
            while True:
                if isinstance(x, tuple):
                    x = g.send([chi2(a) for a in x])
                elif x == math.floor(x):
                    x = g.send(chi2(x))
                else:
                    break                                     # the walk is over: Brent's first iterate
        except StopIteration as stop:
            got = stop.value
        except ValueError:                                    # brentq on a NaN / same-sign bracket, as in the reference
            assert want[0] == 'root'
            continue
        if got is None:
            assert want[0] == 'root', (case, want)
            fr = g.gi_frame.f_locals
            assert (fr['sf_used'], fr['alpha'], fr['alpha0']) == want[1:], (case, want)
        elif got[0] == 'root':                               # ended at once (an end of the bracket is an exact zero)
            assert want[0] == 'root' and got[2]['sf'] == want[1] and got[2]['bracket'] == (want[2], want[3]), (case, want, got)
        else:
            assert got[0] == want[0], (case, want, got)
            if want[0] == 'too_smooth':
                assert got[2]['sf'] == want[1]


def test_refined_search_sees_the_reference_grade_values_only_where_it_matters():
    """With `refine` the walk may come from an approximate evaluator: its values only decide signs (unless within the margin of
    nu), the bracket ends are asked for again as Exact, and Brent runs on the exact function - so a noisy walk gives the
    root of the noise-free search, bit for bit, at the price of two extra evaluations per record."""
    def chi2_true(x):
        return 450. + 300. / (1. + math.exp(-(x + 28.3) * 2.))          # rises through nu = 0.6 * 1000 near -28.3

    calls = dict(plain=0, exact=0)

    def noisy(rec, la, exact=None):
        out = []
        for j, a in enumerate(la):
            e = exact is not None and exact[j]
            calls['exact' if e else 'plain'] += 1
            noise = 0. if (e or a != math.floor(a)) else 2e-5 * math.sin(a * 7.3)
            out.append(chi2_true(a) * (1. + noise))
        return np.array(out)
    clean = AS.run_batched([1000, 1000], lambda r, a: np.array([chi2_true(x) for x in a]), prefetch=8)
    got = AS.run_batched([1000, 1000], noisy, prefetch=8, refine=True)
    assert got[0] == clean[0] and got[1] == clean[1] == ['root', 'root']
    assert calls['exact'] == 4                                           # the two ends of each record's bracket
    # a walk value inside the sign margin of nu is asked for again as well
    calls.update(plain=0, exact=0)
    near = lambda x: 600. * (1. + 2e-4) if x == -10. else chi2_true(x)
    AS.run_batched([1000], lambda r, a, exact=None: (calls.__setitem__('exact', calls['exact'] + int(np.sum(exact)))
                                                    if exact is not None else None) or np.array([near(x) for x in a]),
                   prefetch=8, refine=True)
    assert calls['exact'] == 3


def test_refined_search_redoes_the_walk_when_the_ends_contradict_it():
    """If the exact values at the bracket ends do not change sign (the approximate walk was wrong about one of them), the
    record's walk is redone on exact values - the whole table in one request - and the answer is that of the exact function."""
    def exact_f(x):
        return 450. + 300. / (1. + math.exp(-(x + 28.3) * 2.))

    def approx_f(x):                      # wrong by 30 % at one decade: a spurious sign change between -11 and -12
        return exact_f(x) * (0.7 if x == -12. else 1.)
    seen = []

    def ev(rec, la, exact=None):
        ex = np.zeros(len(la), bool) if exact is None else exact
        seen.append((len(la), int(np.sum(ex))))
        return np.array([exact_f(x) if e or x != math.floor(x) else approx_f(x) for x, e in zip(la, ex)])
    got = AS.run_batched([1000], ev, prefetch=8, refine=True)
    clean = AS.run_batched([1000], lambda r, a: np.array([exact_f(x) for x in a]), prefetch=8)
    assert got[0] == clean[0] and got[2][0]['walk_redone_exact']
    assert (100, 100) in seen                        # the whole table, exact, in one request (the two ends are known)


def test_polish_driver_stops_at_ftol_or_at_a_jump():
    """run_polish_batched: smooth functions end when |f| <= ftol (a few Brent steps), a sign-changing jump ends when it is
    confined to xtol - in about ten rounds, not the ~40 of bisecting to brentq's 2e-12 - and what a record gets does not
    depend on which other records are polished with it."""
    fs = {0: lambda x: (x + 27.3) * 50., 1: lambda x: (-5. if x < -27.123456 else 7.),
          2: lambda x: math.tan((x + 27.5) * 1.2) * 3., 3: lambda x: (x + 27.45)**3 * 1e4 + (x + 27.45)}
    rounds = [0]

    def fb(rec, xs):
        rounds[0] += 1
        return np.array([fs[int(r)](float(x)) for r, x in zip(rec, xs)])
    br = {i: (-27.6, -27.0, fs[i](-27.6), fs[i](-27.0)) for i in fs}
    out = AS.run_polish_batched(br, fb, {i: 1e-3 for i in fs})
    assert rounds[0] <= 12
    assert out[0][3] == 'ftol' and abs(out[0][0] + 27.3) < 1e-4
    assert out[2][3] == 'ftol' and abs(fs[2](out[2][0])) <= 1e-3
    assert out[3][3] == 'ftol' and abs(fs[3](out[3][0])) <= 1e-3
    assert out[1][3] == 'jump' and abs(out[1][0] + 27.123456) <= 1e-7 and abs(out[1][2] - out[1][0]) <= 1e-7
    for i in fs:                                                        # alone: the same answer
        alone = AS.run_polish_batched({i: br[i]}, fb, {i: 1e-3})
        assert alone[i] == out[i], i


def test_vectorised_brent_takes_the_coroutines_steps():
    """BrentBatch (Brent's iteration of all records on arrays) against brentq_gen record by record: same iterates, same
    roots, iteration counts, other ends - on smooth functions, poles, jumps, flat stretches and exact zeros - and
    run_batched gives the same answers with either."""
    rng = np.random.default_rng(11)
    funcs = []
    for k in range(60):
        r, s1 = rng.uniform(-27.9, -27.1), rng.uniform(0.5, 50.)
        kind = k % 6
        if kind == 0:
            funcs.append(lambda x, r=r, s1=s1: s1 * (x - r))
        elif kind == 1:
            funcs.append(lambda x, r=r, s1=s1: s1 * ((x - r)**3 + 1e-3 * (x - r)))
        elif kind == 2:
            funcs.append(lambda x, r=r, s1=s1: -3. if x < r else s1)                         # a jump
        elif kind == 3:
            funcs.append(lambda x, r=r, s1=s1: math.tan((x - r) * 1.4))                       # poles outside the bracket
        elif kind == 4:
            funcs.append(lambda x, r=r, s1=s1: s1 * (x - r) + 0.3 * math.sin(400. * x))       # several roots
        else:
            funcs.append(lambda x, r=r, s1=s1: 0. if abs(x - r) < 1e-3 else s1 * (x - r))     # exact zeros near the root
    n = len(funcs)
    ref = []
    for f in funcs:
        g = AS.brentq_gen(-28., -27., fa=f(-28.), fb=f(-27.))
        xs = []
        try:
            x = next(g)
            while True:
                xs.append(x)
                x = g.send(f(x))
        except StopIteration as stop:
            ref.append((stop.value, xs))
    bb = AS.BrentBatch(n)
    seen = [[] for _ in range(n)]
    for i, f in enumerate(funcs):
        bb.add(i, -28., -27., f(-28.), f(-27.))
    while bb.active.any():
        idx, xs = bb.requests()
        for i, x in zip(idx.tolist(), xs.tolist()):
            seen[i].append(x)
        bb.feed(idx, np.array([funcs[i](x) for i, x in zip(idx.tolist(), xs.tolist())]))
    for i in range(n):
        (root, it, nf, oe), xs = ref[i]
        assert seen[i] == xs, i
        assert bb.results[i] == (root, it, nf, oe), (i, bb.results[i], ref[i][0])
    # the driver, both ways, on chi^2-like functions built from them
    chi = lambda i, a: 600. + funcs[i](a) if -28. <= a <= -27. else (300. if a < -28. else 900.)
    ev = lambda rec, la, exact=None: np.array([chi(int(i), float(a)) for i, a in zip(rec, la)])
    a1 = AS.run_batched([1000] * n, ev, prefetch=8, vector_brent=True)
    a2 = AS.run_batched([1000] * n, ev, prefetch=8, vector_brent=False)
    assert a1[0] == a2[0] and a1[1] == a2[1] and a1[3] == a2[3]
    for i1, i2 in zip(a1[2], a2[2]):
        assert i1 == i2


def test_jump_rule_ends_the_iteration_on_a_jump_and_nowhere_else(monkeypatch):
    """alpha_search.jump_rule (round 4): on a sign change without a root - chi^2 jumping across nu - the iteration ends once the
    bracket is narrower than 1e-7 decades while both ends miss nu by more than 1e-4 nu, instead of bisecting on to brentq's
    2e-12 (40-60 values).  Same answer from the coroutine and from the arrays; the root lies within 1e-7 of the jump; on
    functions WITH a root (smooth, steep, several roots, poles outside) the iterates are those of plain brentq, value for
    value; VINTERP_JUMP_STOP=0 restores brentq's own end."""
    nu = 2600.
    rule = AS.jump_rule(nu)
    assert rule == (1e-7, 1e-4 * nu)
    rng = np.random.default_rng(5)

    def drive_all(f, jump):
        g = AS.brentq_gen(-28., -27., fa=f(-28.), fb=f(-27.), jump=jump)
        xs = []
        try:
            x = next(g)
            while True:
                xs.append(x)
                x = g.send(f(x))
        except StopIteration as stop:
            return stop.value, xs
    jumps, roots = [], []
    for k in range(12):
        r = rng.uniform(-27.9, -27.1)
        jumps.append((r, lambda x, r=r, k=k: (-0.4 - 0.01 * k * (x + 28.)) if x < r else (3.1 + k)))          # |f| >> 0.26 on both sides
        roots.append(lambda x, r=r, k=k: (5. + 40. * k) * nu * (x - r))                                # steep genuine roots
        roots.append(lambda x, r=r, k=k: nu * ((x - r)**3 + 1e-4 * (x - r)))                           # flat genuine roots
        roots.append(lambda x, r=r: math.tan((x - r) * 1.4) * nu)
    for r, f in jumps:
        (root, it, nf, oe), xs = drive_all(f, rule)
        (root0, it0, nf0, oe0), xs0 = drive_all(f, None)
        assert abs(root - r) <= 1e-7 and abs(oe - root) <= 1e-7 and (root - r) * (oe - r) <= 0.
        assert abs(root0 - r) <= 4e-12 and xs == xs0[:len(xs)]                # the same iterates, fewer of them
        assert nf <= nf0 - 10, (nf, nf0)
    for f in roots:
        assert drive_all(f, rule) == drive_all(f, None)
    # the arrays take the coroutine's steps with the rule on
    funcs = [f for _, f in jumps] + roots
    bb = AS.BrentBatch(len(funcs))
    for i, f in enumerate(funcs):
        bb.add(i, -28., -27., f(-28.), f(-27.), jump=rule)
    while bb.active.any():
        idx, xs = bb.requests()
        bb.feed(idx, np.array([funcs[i](x) for i, x in zip(idx.tolist(), xs.tolist())]))
    for i, f in enumerate(funcs):
        assert bb.results[i] == drive_all(f, rule)[0], i
    monkeypatch.setenv('VINTERP_JUMP_STOP', '0')
    assert AS.jump_rule(nu) is None


def test_doubtful_walk_values_are_asked_for_together():
    """A record whose chi^2 sits within the sign margin of a target over a long stretch of the walk (the 50 decades where
    the systems are one and the same matrix, say) asks for their reference-grade values in one request, not decade by
    decade - each request is a round of the evaluator - and ends where the exact function says."""
    def exact_f(x):
        if x <= -45.:
            return 700. * (1. - 3e-3)            # just under nu = 0.7 * 1000: no root at sf = 0.7 but doubtful everywhere
        return 400. + 500. / (1. + math.exp(-(x + 20.) * 1.5))

    rounds = []

    def ev(rec, la, exact=None):
        ex = np.zeros(len(la), bool) if exact is None else np.asarray(exact)
        rounds.append((len(la), int(ex.sum())))
        return np.array([exact_f(x) * (1. if (e or x != math.floor(x)) else 1. + 1e-5 * math.sin(3. * x))
                         for x, e in zip(la, ex)])
    got = AS.run_batched([1000], ev, prefetch=8, refine=True)
    clean = AS.run_batched([1000], lambda r, a: np.array([exact_f(x) for x in a]), prefetch=8)
    assert got[0] == clean[0] and got[1] == clean[1]
    assert sum(1 for _, nx in rounds if nx) <= 3, rounds          # doubtful decades in one or two requests + the bracket ends


def _random_tables(rng, T, npts):
    """Exact and plain (noisy) walk tables of T records: sign changes anywhere, NaNs, exact zeros of chi^2 - nu, values inside
    the sign margin, records that are too smooth at once, plain values that contradict the exact ones."""
    exact = np.empty((T, 102))
    for t in range(T):
        lvl = rng.choice([500., 650., 850., 950., 1050.])
        tab = lvl + rng.choice([1., 30., 300.]) * rng.standard_normal(102)
        if t % 3 == 0:
            tab[rng.integers(1, 102):] -= rng.uniform(100., 600.)
        if t % 5 == 0:
            tab[rng.integers(0, 102)] = float('nan')
        if t % 7 == 0:
            tab[rng.integers(0, 102)] = npts * rng.choice(AS.SCALE_FACTORS)
        if t % 11 == 0:
            tab[rng.integers(0, 102)] = npts * rng.choice(AS.SCALE_FACTORS) * (1. + 3e-4)
        if t % 13 == 0:
            tab[0] = rng.uniform(100., 590.)
        exact[t] = tab
    noise = 1e-5 * rng.standard_normal((T, 102))
    big = rng.random((T, 102)) < 0.01                       # a plain value that is plainly wrong now and then
    noise[big] = 0.2 * rng.standard_normal(int(big.sum()))
    return exact, exact * (1. + noise)


@pytest.mark.parametrize('refine', [False, True])
@pytest.mark.parametrize('prefetch', [1, 8, 32, 102])
def test_table_driver_makes_the_coroutines_requests_and_decisions(refine, prefetch):
    """run_table_batched (the walk of a whole batch on arrays) against run_batched (a coroutine per record): the same requests
    record by record - plain and exact -, the same outcome, scale factor and bracket for every record, on random noisy tables.
    Brent's iteration is replaced by a stub (both drivers hand it the same brackets), so NaN ends are covered too."""
    rng = np.random.default_rng(11 + prefetch)
    T, npts = 300, 1000
    exact, plain = _random_tables(rng, T, npts)
    npl = [npts] * T
    npl[5] = None
    npl[17] = 800

    def make():
        seen = set()
        handed = {}

        def ev(rec, la, ex=None):
            ex = np.zeros(len(la), bool) if ex is None else ex
            out = np.empty(len(la))
            for j, (r, a, e) in enumerate(zip(rec.tolist(), la.tolist(), ex.tolist())):
                assert a == math.floor(a)
                assert (r, a, e) not in seen, 'asked twice'
                seen.add((r, a, e))
                out[j] = (exact if e else plain)[r, int(round(-a))]
            return out

        def solver(ids, brs):
            for i, b in zip(ids, brs):
                handed[i] = dict(b)
            return [(0.5 * (b['alpha'] + b['alpha0']), 3, 2, b['alpha']) for b in brs]
        return seen, handed, ev, solver
    s1, h1, ev1, so1 = make()
    a = AS.run_batched(npl, ev1, prefetch=prefetch, refine=refine, brent_solver=so1)
    s2, h2, ev2, so2 = make()
    b = AS.run_table_batched(npl, ev2, prefetch=prefetch, refine=refine, brent_solver=so2)
    assert s1 == s2, (sorted(s1 - s2)[:5], sorted(s2 - s1)[:5])
    assert a[1] == b[1]
    assert a[3] == b[3]
    assert np.array_equal(np.array(a[0], dtype=float), np.array(b[0], dtype=float), equal_nan=True)
    for i in range(T):
        assert a[2][i] == b[2][i] or (repr(a[2][i]) == repr(b[2][i])), (i, a[2][i], b[2][i])
    assert set(h1) == set(h2)
    for i in h1:
        assert repr(h1[i]) == repr(h2[i]), (i, h1[i], h2[i])
    outc = set(a[1])
    assert {'root', 'no_root', 'too_smooth', 'skipped'} <= outc
    if refine:
        assert any(x.get('walk_redone_exact') for x in a[2])


def test_table_driver_with_brents_iteration():
    """... and with Brent's iteration behind it (BrentBatch on a smooth chi^2 with a noisy walk): the roots of run_batched, bit
    for bit, with and without a solver that declines some records."""
    def chi2_true(r, x):
        return 450. + 300. / (1. + math.exp(-(x + 28.3 + 0.37 * r) * 2.)) + r

    def noisy(rec, la, exact=None):
        out = []
        for j, (r, a) in enumerate(zip(rec.tolist(), la.tolist())):
            e = exact is not None and exact[j]
            noise = 0. if (e or a != math.floor(a)) else 2e-5 * math.sin(a * 7.3 + r)
            out.append(chi2_true(r, a) * (1. + noise))
        return np.array(out)
    npl = [1000] * 40
    ref = AS.run_batched(npl, noisy, prefetch=32, refine=True, vector_brent=True)
    got = AS.run_table_batched(npl, noisy, prefetch=32, refine=True)
    assert got[0] == ref[0] and got[1] == ref[1] and got[2] == ref[2] and got[3] == ref[3]
    assert ref[1].count('root') == 40

    def solver(ids, brs):                       # answers every other record, leaves the rest to the driver
        return [None if i % 2 else (b['alpha'] + 0.25, 4, 3, b['alpha0']) for i, b in zip(ids, brs)]
    ref = AS.run_batched(npl, noisy, prefetch=32, refine=True, brent_solver=solver)
    got = AS.run_table_batched(npl, noisy, prefetch=32, refine=True, brent_solver=solver)
    assert got[0] == ref[0] and got[1] == ref[1] and got[2] == ref[2] and got[3] == ref[3]
