"""Two builds of the library - or one build with and without VINTERP_K3=v1 - on the same systems: eigenvalues (vi_eigvals_f64) and
sweeps bit for bit, and the time per launch.
python tools/ab_jacobi_bits.py before.so after.so [N ...]   (each library is loaded in a process of its own; a path given as
"v1:lib.so" runs that library with VINTERP_K3=v1, i.e. the two-barrier kernel at every order)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np


def run(libpath, N, out):
    lib = C.CDLL(libpath, mode=C.RTLD_GLOBAL)
    VP, I64 = C.c_void_p, C.c_int64
    lib.vi_ctx_create.argtypes = [C.c_int, C.POINTER(VP)]
    lib.vi_dmalloc.argtypes = [VP, C.c_size_t, C.POINTER(VP)]
    lib.vi_h2d.argtypes = [VP, VP, VP, C.c_size_t]
    lib.vi_d2h.argtypes = [VP, VP, VP, C.c_size_t]
    lib.vi_eigvals_f64.argtypes = [VP, I64, C.c_int32, VP, VP, VP]
    lib.vi_timer_start.argtypes = [VP]
    lib.vi_timer_stop_ms.argtypes = [VP, C.POINTER(C.c_double)]
    ctx = VP()
    assert lib.vi_ctx_create(0, C.byref(ctx)) == 0
    rng = np.random.default_rng(0)
    B = 64
    X = np.empty((B, N, N))
    for i in range(B):
        Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
        lam = rng.uniform(0.1, 1., N) if i % 2 else 10.0**rng.uniform(-40, 0, N) * rng.choice([-1, 1], N)
        M = (Q * lam) @ Q.T
        X[i] = 0.5 * (M + M.T)

    def dm(n):
        p = VP()
        assert lib.vi_dmalloc(ctx, n, C.byref(p)) == 0
        return p
    dX, dl, ds = dm(X.nbytes), dm(B * N * 8), dm(B * 4)
    best = 1e9
    for _ in range(3):
        lib.vi_h2d(ctx, dX, X.ctypes.data_as(VP), X.nbytes)
        lib.vi_timer_start(ctx)
        assert lib.vi_eigvals_f64(ctx, B, N, dX, dl, ds) == 0
        ms = C.c_double()
        lib.vi_timer_stop_ms(ctx, C.byref(ms))
        best = min(best, ms.value)
    lam, sw = np.empty((B, N)), np.empty(B, dtype=np.int32)
    lib.vi_d2h(ctx, lam.ctypes.data_as(VP), dl, lam.nbytes)
    lib.vi_d2h(ctx, sw.ctypes.data_as(VP), ds, sw.nbytes)
    np.savez(out, lam=lam, sw=sw, ms=best)


if __name__ == '__main__':
    if sys.argv[1] == '--one':
        run(sys.argv[2], int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    a, b = sys.argv[1], sys.argv[2]
    for N in ([int(x) for x in sys.argv[3:]] if len(sys.argv) > 3 else [144, 32, 12, 100, 160]):
        res = []
        for k, lp in enumerate((a, b)):
            out = '/tmp/ab_%d_%d.npz' % (N, k)
            env = dict(os.environ)
            if lp.startswith('v1:'):
                lp = lp[3:]
                env['VINTERP_K3'] = 'v1'
            subprocess.check_call([sys.executable, os.path.abspath(__file__), '--one', lp, str(N), out], env=env)
            res.append(np.load(out))
        same = np.array_equal(res[0]['lam'], res[1]['lam']) and np.array_equal(res[0]['sw'], res[1]['sw'])
        print('N %3d: eigenvalues and sweeps of 64 systems %s; launch %.3f ms -> %.3f ms (sweeps max %d)'
              % (N, 'IDENTICAL bit for bit' if same else 'DIFFER', float(res[0]['ms']), float(res[1]['ms']), int(res[0]['sw'].max())))
