import os, sys, numpy as np, scipy.linalg, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden, rel
from test_gpu_fit import solve_direct
from volumetricinterp_amd import _lib
f = load_golden('fit_k8l2')
X, y = f['rec0_X'], f['rec0_y']
ref = scipy.linalg.lstsq(X, y)[0]
print('method', os.environ.get('VINTERP_EIG', 'jacobi'))
C, rank, _ = solve_direct(X[None].copy(), y[None])
print('k8l2 rec0 rel', rel(C[0], ref), 'rank', rank)
rng = np.random.default_rng(0)
for N in (9, 32, 33, 144):
    A = rng.standard_normal((3 * N, N)); M = A.T @ A
    D = rng.standard_normal((N, N)); D = D + D.T
    B = 6
    Xb = np.stack([M * 1e-19 + 10.0**(-20 - 2 * i) * D for i in range(B)])
    yb = rng.standard_normal((B, N)) * 1e-8
    Cb, rk, _ = solve_direct(Xb.copy(), yb)
    errs = [rel(Cb[i], scipy.linalg.lstsq(Xb[i], yb[i])[0]) for i in range(B)]
    print('N=%d rank %s max rel err vs lstsq %.2e' % (N, rk, max(errs)))
ctx = _lib.get_context()
A = rng.standard_normal((400, 144)); M = A.T @ A
for B in (1, 8, 102, 256, 1024, 4096):
    Xb = np.stack([M + 10.0**(-(i % 30)) * np.eye(144) for i in range(B)])
    yb = rng.standard_normal((B, 144))
    dX, dy = ctx.to_device(Xb), ctx.to_device(yb)
    dC, drank = ctx.empty((B, 144)), ctx.empty((B,), np.int32)
    eps = np.finfo(float).eps
    def run():
        dX.upload(Xb)
        ctx.timer_start()
        _lib.check(_lib.lib.vi_solve_trunc_f64(ctx.handle, B, 144, dX.ptr, dy.ptr, None, eps, dC.ptr, drank.ptr, 144 * eps, None), 'solve')
        return ctx.timer_stop_ms()
    run(); ms = min(run() for _ in range(3))
    Cb = dC.download()
    r = max(rel(Cb[i], np.linalg.solve(Xb[i], yb[i])) for i in range(min(B, 4)))
    print('B=%d N=144: %.2f ms device, %.1f us/system, err %.1e' % (B, ms, ms * 1e3 / B, r))
