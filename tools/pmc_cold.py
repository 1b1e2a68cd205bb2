"""PMC target: ONE cold truncated solve of a default-order system of the bench record (X = A^T W A + 1e-26.3 R, N = 144)
through vi_solve_trunc_f64 - k_scale_system, k_qr_sim<18> (K3p), k_jacobi_solve<1> (K3), k_qr_back_vec - three times, for
rocprofv3 --pmc passes.   python tools/pmc_cold.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from volumetricinterp_amd import _lib, fitengine  # noqa: F401,E402

f = np.load(os.path.join(ROOT, 'tests', 'golden', 'fit_default_c2.npz'), allow_pickle=True)
X = (f['rec0_AWA'] + 10.0**-26.3 * f['R'])[None]
y = f['rec0_y'][None]
N = 144
ctx = _lib.get_context()
EPS = float(np.finfo(float).eps)
for rep in range(3):
    dX, dy = ctx.to_device(X.copy()), ctx.to_device(y)
    dC, drk = ctx.empty((1, N)), ctx.empty((1,), np.int32)
    _lib.check(_lib.lib.vi_solve_trunc_f64(ctx.handle, 1, N, dX.ptr, dy.ptr, None, EPS, dC.ptr, drk.ptr, N * EPS, None))
    ctx.sync()
print('rank', drk.download())
