"""Experiment: K concurrent fit pipelines (own context / stream / workspace each, one host thread each) on one GPU.
The Brent phase of a batched fit leaves the GPU idle a third of the time (host logic between dependent rounds) and half
empty the rest (a round lasts as long as its slowest system): do independent sub-batches fill the gaps?
Usage (GPU): python tools/exp_streams.py [T] [K]"""
import io, os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volumetricinterp_amd import synth, _lib
from volumetricinterp_amd.fitengine import FitEngine
from volumetricinterp_amd.models.sphharmlag import Model
CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = Model(io.StringIO(CFG)); ctx = m.ctx
lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
P, N = lat.size, m.nbasis
d = [ctx.to_device(a) for a in (lat, lon, alt)]
A = m.basis_device(d[0], d[1], d[2], P, transposed=True).download().T
R = m.eval_reg_matricies['curvature']()
value, error = synth.synth_records(A, T, seed0=1000)
W = error**-2.
ref = None
for K in [int(k) for k in (sys.argv[2:] or ['1', '2', '3', '4'])]:
    bounds = [(T * k) // K for k in range(K + 1)]
    engs = []
    for k in range(K):
        c = _lib.Context(0)
        e = FitEngine(c, c.to_device(np.ascontiguousarray(A.T)), P, N, {'curvature': R}, ['curvature'])
        e.upload_records(W[bounds[k]:bounds[k + 1]], value[bounds[k]:bounds[k + 1]])
        e.fit_resident([P] * (bounds[k + 1] - bounds[k]), calccov=True)          # warm-up (workspace, rocBLAS)
        engs.append(e)
    out = [None] * K

    def run(k):
        out[k] = engs[k].fit_resident([P] * (bounds[k + 1] - bounds[k]), calccov=True)
        engs[k].ctx.sync()
    th = [threading.Thread(target=run, args=(k,)) for k in range(K)]
    t0 = time.perf_counter()
    if os.environ.get('SEQ') == '1':
        for k in range(K):
            run(k)
    else:
        for t in th:
            t.start()
        for t in th:
            t.join()
    dt = time.perf_counter() - t0
    al = np.concatenate([[p['curvature'] for p in o['reg_params']] for o in out])
    if ref is None:
        ref = al
    dl = np.abs(np.log10(al) - np.log10(ref))
    print('K=%d pipelines: %.1f ms -> %.1f records/s; log10 alpha vs K=1: max diff %.2e, records over 1e-6: %d'
          % (K, dt * 1e3, T / dt, np.nanmax(dl), int(np.sum(dl > 1e-6))), flush=True)
    for e in engs:
        e.close()
