"""PMC target: B systems of order 144 through the in-LDS Jacobi kernel alone (vi_eigvals_f64), for rocprofv3 --pmc passes.
python tools/pmc_jacobi.py [B]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import _lib, fitengine  # noqa: F401,E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = 144
ctx = _lib.get_context()
rng = np.random.default_rng(0)
Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
X = (Q * rng.uniform(0.1, 1., N)) @ Q.T
X = np.ascontiguousarray(np.broadcast_to(0.5 * (X + X.T), (B, N, N)))
dX, dl, ds = ctx.to_device(X), ctx.empty((B, N)), ctx.empty((B,), np.int32)
_lib.check(_lib.lib.vi_eigvals_f64(ctx.handle, B, N, dX.ptr, dl.ptr, ds.ptr), 'vi_eigvals_f64')
ctx.sync()
print('B', B, 'sweeps', ds.download()[:4])
