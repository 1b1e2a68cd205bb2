"""Regularisation matrices Omega (curvature) and Psi (0th order) of the sphharmlag model.

Host-side, once per run (not a GPU target).  Restates
``eval_omega/omega_ij`` and ``eval_psi/psi_ij`` of the reference
(volumetricinterp/models/sphharmlag.py:188-239): each entry is a product of three
one-dimensional ``scipy.integrate.quad`` integrals over z, theta and phi.  The integrands and quad
calls are the reference's, so every entry has the reference's value (including the QUADPACK
artefacts of the divergent z-integral of Omega); the only change is that each distinct
1-D integral is evaluated once instead of once per (ni, nj) pair: the z-integral depends only on
(ki, kj), the theta- and phi-integrals only on the ordered pair of (l, m) indices - 10 + 2 x 1 296
quadratures at the default order instead of 31 320.
"""
import os
import warnings

import numpy as np
import scipy.integrate
import scipy.special as sp


# ---- the theta- and phi-integrals of one (l, m) pair as a task a worker process can run -------------------------------------
# (module-level, picklable; the same scipy / numpy calls on the same scalars as the closures they replace, so a value does
# not depend on which process computed it)
def _kvm(v, m):                                 # sphharmlag.py:305-321
    with np.errstate(all='ignore'):
        K = np.sqrt((2 * v + 1) / (4 * np.pi) * sp.gamma(float(v - m + 1)) / sp.gamma(float(v + m + 1)))
    if m != 0:
        K = K * np.sqrt(2)
    return K


def _az(v, m, phi):                             # sphharmlag.py:263-281
    if m < 0:
        return _kvm(v, abs(m)) * np.sin(abs(m) * phi)
    return _kvm(v, abs(m)) * np.cos(abs(m) * phi)


def _lap(m, v, t):                              # sphharmlag.py:201-206
    c = np.cos(t)
    return (-1 * v * (v * c**2 + v + 1) * sp.lpmv(m, v, c) + v * (v + m) * c * sp.lpmv(m, v - 1, c)
            + v * (v - m + 1) * c * sp.lpmv(m, v + 1, c))


def _t_integrand(kind, mi, vi, mj, vj):
    if kind == 'omega':
        return lambda t: 1 / np.sin(t)**3 * _lap(mi, vi, t) * _lap(mj, vj, t)
    return lambda t: sp.lpmv(mi, vi, np.cos(t)) * sp.lpmv(mj, vj, np.cos(t)) * np.sin(t)


def _tp_task(task):
    kind, mi, vi, mj, vj, cap = task
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')          # IntegrationWarning: the reference emits (and ignores) them too
        It = scipy.integrate.quad(_t_integrand(kind, mi, vi, mj, vj), 0., cap)[0]
        Ip = scipy.integrate.quad(lambda p: _az(vi, mi, p) * _az(vj, mj, p), 0., 2 * np.pi)[0]
    return It, Ip


def _tp_chunk(tasks):
    return [_tp_task(t) for t in tasks]


def _tp_parallel(tasks, nw):
    """The tasks over nw worker processes: plain child processes running THIS FILE as a script (numbers in and out through
    .npy files in a private temporary directory - nothing is pickled, nothing of the parent - GPU runtime, __main__ - is
    inherited or re-imported; multiprocessing's spawn re-imports the parent's main module, which hangs when that is an
    interactive session or a script on stdin).  Worker i takes tasks i, i + nw, ...: equal shares of cheap and dear pairs."""
    import subprocess
    import sys
    import tempfile
    kinds = {'omega': 0., 'psi': 1.}
    arr = np.array([[kinds[t[0]]] + [float(x) for x in t[1:]] for t in tasks], dtype=np.float64)
    with tempfile.TemporaryDirectory(prefix='vinterp_regmat_') as d:
        np.save(os.path.join(d, 'tasks.npy'), arr)
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--worker', d, str(i), str(nw)], env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for i in range(nw)]
        errs = [p_.communicate()[1] for p_ in procs]
        for p_, e in zip(procs, errs):
            if p_.returncode != 0:
                raise RuntimeError('regmat worker failed (status %d): %s' % (p_.returncode, e.decode('utf-8', 'replace')[-2000:]))
        out = np.empty((len(tasks), 2))
        for i in range(nw):
            out[i::nw] = np.load(os.path.join(d, 'out_%d.npy' % i))
    return [(float(a), float(b)) for a, b in out]


def _worker_main(argv):
    d, i, nw = argv[0], int(argv[1]), int(argv[2])
    arr = np.load(os.path.join(d, 'tasks.npy'))[i::nw]
    names = {0.: 'omega', 1.: 'psi'}
    res = _tp_chunk([(names[r[0]], r[1], r[2], r[3], r[4], r[5]) for r in arr])
    np.save(os.path.join(d, 'out_%d.npy' % i), np.array(res, dtype=np.float64).reshape(len(res), 2))


PARALLEL_MIN_TASKS = 3000       # distinct (l, m) pairs from which the angular integrals go to worker processes


def _workers(ntasks):
    env = os.environ.get('VINTERP_REGMAT_WORKERS')
    if env is not None:
        return max(1, int(env))
    if ntasks < PARALLEL_MIN_TASKS:
        return 1
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:                      # pragma: no cover
        ncpu = os.cpu_count() or 1
    return max(1, min(16, ncpu))


def _assemble(model, z_int, kind):
    """Entries = z-integral(ki, kj) x theta-integral x phi-integral of the ordered (l, m) pair, each distinct 1-D integral once.
    At the doubled order of BASELINE configs[4] (MAXK 8 x MAXL 12, N = 1152) there are 20 736 distinct angular pairs - 65 s of
    QUADPACK calls in one process (measured, round 4; the reference's own loop would take ~25 min) - so from
    PARALLEL_MIN_TASKS pairs on they are spread over worker processes (spawned: nothing of a GPU runtime is inherited); the
    values are the same bit for bit (tests/test_host_tables.py).  VINTERP_REGMAT_WORKERS=1 keeps everything in this process."""
    N = model.nbasis
    L2 = model.maxl**2
    out = np.zeros((N, N))
    klm = [model.basis_numbers(n) for n in range(N)]
    nus = [model.nu(n) for n in range(N)]
    cap = model.cap_lim
    # distinct angular pairs, in the order the serial loop meets them
    keys, tasks = {}, []
    for ni in range(N):
        for nj in range(ni, N):
            kt = (ni % L2, nj % L2)
            if kt not in keys:
                keys[kt] = len(tasks)
                tasks.append((kind, klm[ni][2], nus[ni], klm[nj][2], nus[nj], cap))
    nw = _workers(len(tasks))
    vals = _tp_parallel(tasks, nw) if nw > 1 else _tp_chunk(tasks)
    Iz_cache = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for ni in range(N):
            ki = klm[ni][0]
            for nj in range(ni, N):
                kj = klm[nj][0]
                kz = (ki, kj)
                if kz not in Iz_cache:
                    Iz_cache[kz] = scipy.integrate.quad(z_int(ki, kj), 0., model.max_z_int)[0]
                It, Ip = vals[keys[(ni % L2, nj % L2)]]
                v = Iz_cache[kz] * It * Ip
                out[ni, nj] = v
                out[nj, ni] = v
    return out


# sphharmlag.py:198-212
def eval_omega(model):
    def z_int(ki, kj):
        return lambda z: np.exp(-1 * z) * sp.eval_laguerre(ki, z) * sp.eval_laguerre(kj, z) / z**2
    return _assemble(model, z_int, 'omega')


# sphharmlag.py:224-239
def eval_psi(model):
    def z_int(ki, kj):
        return lambda z: np.exp(-1 * z) * sp.eval_laguerre(ki, z) * sp.eval_laguerre(kj, z) * z**2
    return _assemble(model, z_int, 'psi')


# ---- Psi by Gauss quadrature -----------------------------------------------------------------------
def eval_psi_gauss(model, ntheta=None):
    """Psi (sphharmlag.py:215-239) from closed-form / Gaussian quadrature instead of 3 x N(N+1)/2 adaptive QUADPACK calls.

    The three factors of psi_ij are convergent and smooth (unlike Omega's z-integral, which diverges - SURVEY F5 - so that
    the reference's Omega is a QUADPACK artefact no other quadrature can reproduce):
      z      int_0^inf e^-z L_ki L_kj z^2 dz   - a polynomial of degree <= 2 maxk against e^-z: Gauss-Laguerre with maxk + 2
             nodes is exact (finite MAX_Z_INT: the reference's adaptive quad call, once per (ki, kj));
      theta  int_0^cap P_vi^mi(cos t) P_vj^mj(cos t) sin t dt  - Gauss-Legendre on [0, cap], nodes growing with the degree;
      phi    int_0^2pi Az_i Az_j dphi = Kvm_i Kvm_j x (2 pi | pi | 0)  - analytic.
    One table per factor, the matrix is their outer combination: milliseconds at N = 144 (2.4 s with the de-duplicated
    QUADPACK calls of eval_psi, 10 s in the reference), 0.2 s at N = 1152 (52 s / ~25 min).  Agrees with the reference's
    own Psi to 1e-10 of max|Psi| (tests/test_host_tables.py) - QUADPACK's tolerance, not this routine's."""
    N, maxk, maxl = model.nbasis, model.maxk, model.maxl
    L2 = maxl**2
    klm = [model.basis_numbers(n) for n in range(L2)]
    nus = np.array([model.nu(n) for n in range(L2)], dtype=np.float64)
    ms = np.array([klm[n][2] for n in range(L2)], dtype=np.float64)
    # z
    if np.isinf(model.max_z_int):
        zx, zw = np.polynomial.laguerre.laggauss(maxk + 2)
        zw = zw * zx**2
        Lk = np.array([sp.eval_laguerre(k, zx) for k in range(maxk)])
        Iz = (Lk * zw) @ Lk.T
    else:
        # finite limit: e^-z z^2 L_ki L_kj on [0, Z] is no polynomial weight any more and a fixed Gauss-Legendre rule is only
        # converged while Z is small against its node count (a limit of 1e3 used as 'practically infinite' is not).  There
        # are only maxk (maxk + 1) / 2 such integrals: the reference's own adaptive quad call for each (sphharmlag.py:230).
        Iz = np.zeros((maxk, maxk))
        for ki in range(maxk):
            for kj in range(ki, maxk):
                Iz[ki, kj] = Iz[kj, ki] = scipy.integrate.quad(
                    lambda z: np.exp(-1 * z) * sp.eval_laguerre(ki, z) * sp.eval_laguerre(kj, z) * z**2, 0., model.max_z_int)[0]
    Iz = 0.5 * (Iz + Iz.T)                      # the products round asymmetrically; the reference fills Psi symmetrically
    # theta: signed order into lpmv, as the reference does
    if ntheta is None:
        ntheta = int(max(96, 4 * np.ceil(np.max(nus)) + 32))
    gx, gw = np.polynomial.legendre.leggauss(ntheta)
    t = 0.5 * model.cap_lim * (gx + 1.)
    tw = 0.5 * model.cap_lim * gw * np.sin(t)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Pm = np.array([sp.lpmv(ms[r], nus[r], np.cos(t)) for r in range(L2)])
    It = (Pm * tw) @ Pm.T
    It = 0.5 * (It + It.T)
    # phi: Az(v, m, p) = Kvm(v, |m|) * (sin(|m| p) if m < 0 else cos(|m| p))   (sphharmlag.py:263-281)
    K = np.array([model.Kvm(nus[r], abs(ms[r])) for r in range(L2)])
    same = ms[:, None] == ms[None, :]
    Ip = np.where(same, np.where(ms[:, None] == 0, 2. * np.pi, np.pi), 0.) * np.outer(K, K)
    ang = It * Ip
    kk = np.arange(N) // L2
    rr = np.arange(N) % L2
    return Iz[np.ix_(kk, kk)] * ang[np.ix_(rr, rr)]


if __name__ == '__main__':
    import sys
    if len(sys.argv) == 5 and sys.argv[1] == '--worker':
        _worker_main(sys.argv[2:])
