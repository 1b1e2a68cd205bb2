"""Where k_qr_sim spends its cycles (stamped diagnostic build csrc/libvinterp_st.so: vi_qr.hip compiled with -DVI_STAMPS):
cycle sums of thread 0 of workgroup 0 for one 144 x 144 system.  python tools/exp_qr_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, 'volumetricinterp_amd', 'csrc', 'libvinterp_st.so'), mode=C.RTLD_GLOBAL)
VP, I64 = C.c_void_p, C.c_int64
lib.vi_ctx_create.argtypes = [C.c_int, C.POINTER(VP)]
lib.vi_dmalloc.argtypes = [VP, C.c_size_t, C.POINTER(VP)]
lib.vi_h2d.argtypes = [VP, VP, VP, C.c_size_t]
lib.vi_qr_similarity_f64.argtypes = [VP, I64, C.c_int32, VP, VP, VP, VP, VP]
lib.vi_debug_qr_stamps.argtypes = [C.POINTER(C.c_double), C.c_int]
lib.vi_ctx_sync.argtypes = [VP]
lib.vi_last_error.restype = C.c_char_p
ctx = VP()
assert lib.vi_ctx_create(0, C.byref(ctx)) == 0


def dmalloc(n):
    p = VP()
    assert lib.vi_dmalloc(ctx, n, C.byref(p)) == 0
    return p


f = np.load(os.path.join(ROOT, 'tests', 'golden', 'fit_default_c2.npz'), allow_pickle=True)
X = f['rec0_AWA'] + 10.0**-26.3 * f['R']
X = X * 2.0**(1 - np.frexp(np.max(np.abs(X)))[1])
N = X.shape[0]
y = np.ascontiguousarray(f['rec0_y'])
dX, dy, dX1, dy1, dQ = dmalloc(X.nbytes), dmalloc(y.nbytes), dmalloc(X.nbytes), dmalloc(y.nbytes), dmalloc(X.nbytes)
lib.vi_h2d(ctx, dX, X.ctypes.data_as(VP), X.nbytes)
lib.vi_h2d(ctx, dy, y.ctypes.data_as(VP), y.nbytes)
out = (C.c_double * 8)()
for rep in range(3):
    lib.vi_debug_qr_stamps(out, 1)
    assert lib.vi_qr_similarity_f64(ctx, 1, N, dX, dy, dX1, dy1, dQ) == 0, lib.vi_last_error()
    lib.vi_ctx_sync(ctx)
    lib.vi_debug_qr_stamps(out, 0)
    v = list(out)
    names = ['load + init', 'pivot search + barrier (sum)', 'reflector by the owner + barrier (sum)', 'update pass (sum)',
             'tail stores + barrier', 'phase 2 load', 'phase 2 reflections', '-']
    print('run %d: total %.0f cycles' % (rep, sum(v)))
    for n_, c_ in zip(names, v):
        print('   %-42s %10.0f cycles  (%.0f per step)' % (n_, c_, c_ / (N - 1)))
