"""Rotations per sweep of the K3 launches of a batched fit (diagnostic build tools/microbench/libvinterp_rothist.so: vi_jacobi.hip
compiled with -DVI_ROTHIST): by sweep index, the solves that ran that sweep and the pairs they rotated in it (of N(N-1)/2 =
10296 at N = 144).  The device-side Brent kernel is not counted (its solves live in vi_brent.hip).
    python tools/exp_rot_hist.py [T]"""
import ctypes as C
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('VINTERP_LIB', os.path.join(ROOT, 'tools', 'microbench', 'libvinterp_rothist.so'))
os.environ.setdefault('VINTERP_PIPELINES', '1')
from volumetricinterp_amd import synth, _lib                      # noqa: E402
from volumetricinterp_amd.fitengine import FitEngine              # noqa: E402
from volumetricinterp_amd.models.sphharmlag import Model          # noqa: E402

CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = Model(io.StringIO(CFG))
ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
P, N = lat.size, m.nbasis
d = [ctx.to_device(a) for a in (lat, lon, alt)]
At = m.basis_device(d[0], d[1], d[2], P, transposed=True)
A = At.download().T
R = m.eval_reg_matricies['curvature']()
value, error = synth.synth_records(A, T, seed0=1000)
eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
eng.upload_records(error**-2., value)
out = (C.c_double * 64)()
_lib.lib.vi_debug_rot_hist.argtypes = [C.POINTER(C.c_double), C.c_int]
_lib.lib.vi_debug_rot_hist(out, 1)
eng.fit_resident([P] * T)
ctx.sync()
_lib.lib.vi_debug_rot_hist(out, 0)
h = np.array(list(out)).reshape(32, 2)
print('sweep: solves that ran it, rotations per solve in it')
for i in range(32):
    if h[i, 1] > 0:
        print('  %2d: %7d  %8.1f' % (i + 1, h[i, 1], h[i, 0] / h[i, 1]))
print('sweeps in all: %d' % h[:, 1].sum())
