import os, sys, numpy as np, scipy.linalg, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_golden, rel
from test_gpu_fit import solve_direct
f = load_golden('fit_k8l2')
X, y = f['rec0_X'], f['rec0_y']
ref = scipy.linalg.lstsq(X, y)[0]
print('method', os.environ.get('VINTERP_EIG', 'syevd'))
C, rank, _ = solve_direct(X[None].copy(), y[None])
print('raw     rel', rel(C[0], ref), 'rank', rank)
s = np.abs(X).max()
C, rank, _ = solve_direct((X / s)[None].copy(), (y / s)[None])
print('scaled  rel', rel(C[0], ref), 'rank', rank)
lam = np.linalg.eigvalsh(X)
print('eig range', lam.min(), lam.max(), 'min abs', np.abs(lam).min())
# timing of batched solves, N=144
g = load_golden('fit_default')
R = g['R']; 
rng = np.random.default_rng(0)
A = rng.standard_normal((400, 144)); M = A.T @ A
for B in (1, 16, 128, 1024):
    Xb = np.stack([M + 10.0**(-i % 30) * np.eye(144) for i in range(B)])
    yb = rng.standard_normal((B, 144))
    solve_direct(Xb.copy(), yb)
    t = time.time(); Cb, rk, _ = solve_direct(Xb.copy(), yb); dt = time.time() - t
    r = max(rel(Cb[i], np.linalg.solve(Xb[i], yb[i])) for i in range(min(B, 4)))
    print('B=%d N=144: %.1f ms total (incl. transfers), %.3f ms/system, err %.1e' % (B, dt * 1e3, dt * 1e3 / B, r))
