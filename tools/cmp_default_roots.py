"""GPU fit of the default-order fixture records against the reference's root set (tests/golden/fit_default_roots.npz: 16
perturbed runs with SciPy's default LAPACK driver gelsd; fit_default_drivers.npz: 8 runs each with gelss and gelsy;
fit_default{16,_c2}.npz: 3 gelsd runs).  Per record: the GPU's root, the distance to the nearest root of each driver's
runs, the density deviation from the nearest run.   python tools/cmp_default_roots.py"""
import math
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from volumetricinterp_amd.interpolate import Interpolate          # noqa: E402
from volumetricinterp_amd.estimate import Estimate                # noqa: E402
from volumetricinterp_amd import synth                            # noqa: E402

G = os.path.join(ROOT, 'tests', 'golden')
r, d = np.load(os.path.join(G, 'fit_default_roots.npz')), np.load(os.path.join(G, 'fit_default_drivers.npz'))
rel = lambda x, y: float(np.linalg.norm(x - y) / np.linalg.norm(y))      # noqa: E731
for tag in ('default_c2', 'default16'):
    f = np.load(os.path.join(G, 'fit_%s.npz' % tag), allow_pickle=True)
    T = f['value'].shape[0]
    with tempfile.TemporaryDirectory() as td:
        cfg = os.path.join(td, 'c.ini')
        open(cfg, 'w').write(str(f['cfg']))
        it = Interpolate(cfg)
        res = it.fit_records(f['lat'], f['lon'], f['alt'], f['value'], f['error'], {'curvature': f['R']})
    es = Estimate.from_arrays(np.nan_to_num(res['Coeffs']), None, f['utime'], f['hull_vert'], str(f['cfg']))
    g = synth.query_grid(8)
    dens = [es.evaluate_coeffs(np.nan_to_num(res['Coeffs'][t:t + 1]), *g, check_hull=True)[0].ravel() for t in range(T)]
    runs = {'gelsd': (np.concatenate([[f['alpha'], f['alpha_p1'], f['alpha_p2']], r[tag + '_alpha']]),
                      np.concatenate([[f['dens'], f['dens_p1'], f['dens_p2']], r[tag + '_dens']])),
            'gelss': (d[tag + '_gelss_alpha'], d[tag + '_gelss_dens']), 'gelsy': (d[tag + '_gelsy_alpha'], d[tag + '_gelsy_dens'])}
    info = res['search']['curvature']['info']
    for t in range(T):
        a = res['reg_params'][t]['curvature']
        if not (a > 0):
            print('%s rec %2d: GPU outcome %s' % (tag, t, res['search']['curvature']['outcomes'][t]))
            continue
        la = math.log10(a)
        line = '%s rec %2d: GPU %.4f sf %.1f chi2-nu %+.1e%s |' % (tag, t, la, info[t]['sf'], info[t].get('chi2_minus_nu', float('nan')),
                                                               ' jump' if info[t].get('jump') else '')
        for drv, (al, dn) in runs.items():
            lr = np.log10(np.where(al[:, t] > 0, al[:, t], np.nan))
            k = int(np.nanargmin(np.abs(lr - la)))
            ok = np.isfinite(dn[k, t].ravel())
            same = np.abs(lr - lr[k]) <= 0.02
            spread_a = float(np.nanmax(lr[same]) - np.nanmin(lr[same]))
            spread_d = max([rel(dn[j, t].ravel()[ok], dn[k, t].ravel()[ok]) for j in np.nonzero(same)[0] if j != k] or [0.])
            line += ' %s: d %.1e (n %d, spread %.0e) dens %.1e (spread %.0e) |' % (
                drv, abs(lr[k] - la), int(same.sum()), spread_a, rel(dens[t][ok], dn[k, t].ravel()[ok]), spread_d)
        print(line)
