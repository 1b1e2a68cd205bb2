"""Microbenchmark of the in-LDS Jacobi eigen-solve kernel alone (vi_eigvals_f64): B systems of order N, time per
launch by HIP events, reported per system, per sweep and as LDS traffic.  Loads the library given on the command
line directly through ctypes (so that two builds can be compared in one call):
    python tools/perf_jacobi.py [lib.so] [N] """
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libpath = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'volumetricinterp_amd', 'csrc', 'libvinterp.so')
N = int(sys.argv[2]) if len(sys.argv) > 2 else 144
lib = C.CDLL(libpath, mode=C.RTLD_GLOBAL)
VP, I64 = C.c_void_p, C.c_int64
lib.vi_ctx_create.argtypes = [C.c_int, C.POINTER(VP)]
lib.vi_dmalloc.argtypes = [VP, C.c_size_t, C.POINTER(VP)]
lib.vi_h2d.argtypes = [VP, VP, VP, C.c_size_t]
lib.vi_d2h.argtypes = [VP, VP, VP, C.c_size_t]
lib.vi_eigvals_f64.argtypes = [VP, I64, C.c_int32, VP, VP, VP]
lib.vi_timer_start.argtypes = [VP]
lib.vi_timer_stop_ms.argtypes = [VP, C.POINTER(C.c_double)]
lib.vi_last_error.restype = C.c_char_p
ctx = VP()
assert lib.vi_ctx_create(0, C.byref(ctx)) == 0, lib.vi_last_error()


def dmalloc(nbytes):
    p = VP()
    assert lib.vi_dmalloc(ctx, nbytes, C.byref(p)) == 0
    return p


rng = np.random.default_rng(0)
print('library', os.path.basename(libpath), 'N', N)
for kind in ('well-conditioned', 'graded rank-deficient'):
    for B in (1, 256, 1024):
        X = np.empty((B, N, N))
        for i in range(min(B, 8)):
            Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
            if kind == 'well-conditioned':
                lam = rng.uniform(0.1, 1., N)
            else:
                lam = 10.0**rng.uniform(-40, 0, N) * rng.choice([-1, 1], N)
            M = (Q * lam) @ Q.T
            X[i] = 0.5 * (M + M.T)
        for i in range(8, B):
            X[i] = X[i % 8]
        dX, dl, ds = dmalloc(X.nbytes), dmalloc(B * N * 8), dmalloc(B * 4)
        best = 1e30
        for rep in range(3):
            lib.vi_h2d(ctx, dX, X.ctypes.data_as(VP), X.nbytes)
            lib.vi_timer_start(ctx)
            rc = lib.vi_eigvals_f64(ctx, B, N, dX, dl, ds)
            assert rc == 0, lib.vi_last_error()
            ms = C.c_double()
            lib.vi_timer_stop_ms(ctx, C.byref(ms))
            best = min(best, ms.value)
        sw = np.empty(B, dtype=np.int32)
        lib.vi_d2h(ctx, sw.ctypes.data_as(VP), ds, sw.nbytes)
        tot_sw = float(sw.sum())
        lds = 2 * 8 * (N * (N + 1) // 2)          # bytes per pass of the matrix
        print('%-22s B=%5d: %8.3f ms  %7.2f us/system  sweeps mean %.1f max %d  %.1f us per sweep of the slowest system'
              % (kind, B, best, best * 1e3 / B, sw.mean(), sw.max(), best * 1e3 / sw.max()))
