"""Diagnostic (STAMPS=1 build): where a Jacobi round spends its cycles (wave 0 of workgroup 0)."""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from volumetricinterp_amd import _lib, fitengine
ctx = _lib.get_context()
rng = np.random.default_rng(0)
N = 144
A = rng.standard_normal((400, N)); M = A.T @ A
X = np.stack([M + np.eye(N)])
dX = ctx.to_device(X); dl = ctx.empty((1, N)); ds = ctx.empty((1,), np.int32)
out = (C.c_double * 8)()
_lib.lib.vi_debug_jacobi_stamps(out, 1)
_lib.check(_lib.lib.vi_eigvals_f64(ctx.handle, 1, N, dX.ptr, dl.ptr, ds.ptr), 'eig'); ctx.sync()
_lib.lib.vi_debug_jacobi_stamps(out, 1)
sw = int(ds.download()[0]); rounds = sw * (N - 1)
names = ['phase1 (rotation set-up)', 'phase2a block fetch', 'barrier 1 wait', 'phase2b update+store', 'barrier 2 wait',
         'truncate/reduce', 'replay (all rounds)', 'loop overhead']
print('sweeps', sw, 'rounds', rounds)
tot = sum(out)
for n, v in zip(names, out):
    print('%-28s %10.0f ticks  %6.1f per round  %5.1f%%' % (n, v, v / rounds, 100 * v / tot))
