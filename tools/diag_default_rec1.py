"""Diagnostic (GPU): why the search's chi^2 and the final chi^2 disagree on golden fit_default record 1.

For each record of tests/golden/fit_default.npz: run the fit, then evaluate chi^2 at the returned root and at the
bracket ends three ways - cold (vi_solve_trunc_f64 on the untransformed system), warm (the record's rotated system,
as the Brent iterates were served) and, with VINTERP_EIG=syevd in the environment, through rocSOLVER.
Usage: python tools/diag_default_rec1.py [fixture]      (env VINTERP_WARM / VINTERP_EIG / VINTERP_ROOT honoured)"""
import math
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden, rel                      # noqa: E402
from test_gpu_fit import make_interp, reg_of               # noqa: E402
from volumetricinterp_amd.fitengine import FitEngine       # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'fit_default'
f = load_golden(name)
regm, reg = reg_of(f)
it = make_interp(tempfile.mkdtemp(), str(f['cfg']))
ctx = it.ctx
lat, lon, alt = f['lat'], f['lon'], f['alt']
P = lat.size
it.model.handle(ctx)
At = it.model.basis_device(ctx.to_device(lat), ctx.to_device(lon), ctx.to_device(alt), P, transposed=True)
fin = np.isfinite(f['value'])
W = np.where(fin, f['error']**-2., 0.)
b = np.where(fin, f['value'], 0.)
T = W.shape[0]
npts = [int(x) for x in fin.sum(1)]
print('env: WARM=%s EIG=%s ROOT=%s' % tuple(os.environ.get(k, '-') for k in ('VINTERP_WARM', 'VINTERP_EIG', 'VINTERP_ROOT')))
eng = FitEngine(ctx, At, P, it.model.nbasis, regm, [reg])
res = eng.fit(W, b, npts)
for t in range(T):
    info = res['search'][reg]['info'][t]
    oc = res['search'][reg]['outcomes'][t]
    a = res['reg_params'][t][reg]
    print('record %d: outcome %s  sf %s  bracket %s  root %s  finder %s iters %s | final chi2 %.6f  nu %.1f  ref alpha %.6f ref chi2 %.6f'
          % (t, oc, info.get('sf'), info.get('bracket'), info.get('log10_alpha'), info.get('finder'), info.get('iterations'),
             res['chi_sq'][t], (info.get('sf') or float('nan')) * npts[t], math.log10(f['alpha'][t]) if f['alpha'][t] > 0 else float('nan'),
             f['chi_sq'][t]))
    if oc != 'root':
        continue
    root = info['log10_alpha']
    lo, hi = info['bracket']
    xs = np.array([lo, hi, root, root - 1e-9, root + 1e-9, root - 1e-4, root + 1e-4, 0.5 * (lo + hi)])
    rec = np.full(len(xs), t, dtype=np.int32)
    cold = eng.chi2_batch(rec, {reg: np.power(10., xs)})
    # warm: through the search entry (non-integer requests use the rotated system set up by the search of the
    # LAST record-set; re-create it at the root so that the basis is the one Brent would have had at its first iterate)
    eng._warm_reset()
    warm = eng.chi2_batch_search(rec, np.where(xs == np.floor(xs), xs + 1e-13, xs), reg) if eng.warm_enabled() else cold * np.nan
    for x, c, w in zip(xs, cold, warm):
        print('   log10a %+.12f   cold chi2 %.9f   warm chi2 %.9f   diff %.3e' % (x, c, w, w - c))
# chi^2 table on the integer walk vs the reference's own values
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_alpha_search import _split_calls                  # noqa: E402
tables = _split_calls(f['chi2_calls'], npts)
for t, tab in enumerate(tables):
    al = np.array(sorted([a for a in tab if a == int(a)], reverse=True))
    mine = eng.chi2_batch(np.full(len(al), t, dtype=np.int32), {reg: np.power(10., al)})
    print('record', t, 'min chi2 over the walk: gpu %.4f at %g | reference %.4f' % (mine.min(), al[mine.argmin()], min(tab[a] for a in al)))
    for a, m in zip(al, mine):
        if -50 <= a <= -20:
            print('  a=%5.0f  ref chi2 %.9g   gpu chi2 %.9g   rel diff %.1e' % (a, tab[a], m, abs(m - tab[a]) / abs(tab[a])))
