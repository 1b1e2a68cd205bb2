"""Diagnostic: chi^2(alpha) table of the GPU path vs the values the reference evaluated (fixture chi2_calls)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden, rel
from test_alpha_search import _split_calls
from test_gpu_fit import make_interp, reg_of
import tempfile
name = sys.argv[1] if len(sys.argv) > 1 else 'fit_default'
f = load_golden(name)
regm, reg = reg_of(f)
class TP:  # tmp_path stand-in
    def __init__(s): s.d = tempfile.mkdtemp()
    def __str__(s): return s.d
it = make_interp(TP(), str(f['cfg']))
from volumetricinterp_amd.fitengine import FitEngine
ctx = it.ctx
lat, lon, alt = f['lat'], f['lon'], f['alt']
P = lat.size
it.model.handle(ctx)
At = it.model.basis_device(ctx.to_device(lat), ctx.to_device(lon), ctx.to_device(alt), P, transposed=True)
fin = np.isfinite(f['value'])
W = np.where(fin, f['error']**-2., 0.); b = np.where(fin, f['value'], 0.)
eng = FitEngine(ctx, At, P, it.model.nbasis, regm, [reg])
eng.load_records(W, b)
npts = [int(x) for x in fin.sum(1)]
tables = _split_calls(f['chi2_calls'], npts)
for t, tab in enumerate(tables):
    al = np.array(sorted([a for a in tab if a == int(a)], reverse=True))
    mine = eng.chi2_batch(np.full(len(al), t, dtype=np.int32), {reg: np.power(10., al)})
    print('record', t, 'npts', npts[t], 'ref alpha', f['alpha'][t])
    for a, m in zip(al, mine):
        if a > -60 or a % 10 == 0:
            print('  a=%5.0f  ref chi2 %.9g   gpu chi2 %.9g   rel diff %.1e' % (a, tab[a], m, abs(m - tab[a]) / abs(tab[a])))
