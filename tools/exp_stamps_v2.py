"""Diagnostic (GPU, stamped build tools/microbench/libvinterp_stamps.so = make with -DVI_STAMPS): where a Jacobi round
spends its cycles.  Segments, per wave 0 (which computes the rotations) and the last wave (which only updates blocks):
0 phase 1 (rotations), 1 block fetch issue, 2 wait at barrier 1, 3 rotation reads + update + stores, 4 wait at barrier 2,
7 loop overhead.  Shares only - the stamped build itself runs slower than the product kernel."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, 'tools', 'microbench', 'libvinterp_stamps.so'), mode=C.RTLD_GLOBAL)
VP, I64 = C.c_void_p, C.c_int64
lib.vi_ctx_create.argtypes = [C.c_int, C.POINTER(VP)]
lib.vi_dmalloc.argtypes = [VP, C.c_size_t, C.POINTER(VP)]
lib.vi_h2d.argtypes = [VP, VP, VP, C.c_size_t]
lib.vi_d2h.argtypes = [VP, VP, VP, C.c_size_t]
lib.vi_eigvals_f64.argtypes = [VP, I64, C.c_int32, VP, VP, VP]
lib.vi_ctx_sync.argtypes = [VP]
ctx = VP()
assert lib.vi_ctx_create(0, C.byref(ctx)) == 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 144
rng = np.random.default_rng(0)
Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
X = (Q * rng.uniform(0.1, 1., N)) @ Q.T
X = np.ascontiguousarray(0.5 * (X + X.T))[None]


def dmalloc(n):
    p = VP()
    assert lib.vi_dmalloc(ctx, n, C.byref(p)) == 0
    return p


dX, dl, ds = dmalloc(X.nbytes), dmalloc(N * 8), dmalloc(4)
out = (C.c_double * 16)()
lib.vi_debug_jacobi_stamps(out, 1)
lib.vi_h2d(ctx, dX, X.ctypes.data_as(VP), X.nbytes)
assert lib.vi_eigvals_f64(ctx, 1, N, dX, dl, ds) == 0
lib.vi_ctx_sync(ctx)
sw = np.zeros(1, dtype=np.int32)
lib.vi_d2h(ctx, sw.ctypes.data_as(VP), ds, 4)
lib.vi_debug_jacobi_stamps(out, 0)
v = np.array(list(out))
Np = (N + 3) & ~3
rounds = sw[0] * Np // 2
names = ['s0 set-up | wait stores+fetch', 's1 barrier', 's2 diag copies | update+post', 's3 wait mailbox | wait go', 's4 read+go+DPP | stores issued', 's5', 's6', 's7 signal+loop']
print('N %d sweeps %d rounds %d' % (N, sw[0], rounds))
for w, off in (('wave 0', 0), ('last wave', 8)):
    tot = v[off:off + 8].sum()
    print(w, 'cycles per round %.0f:' % (tot / rounds), '  '.join('%s %.0f' % (names[k], v[off + k] / rounds) for k in (0, 1, 2, 3, 4, 7)))
