#!/usr/bin/env python3
"""Benchmark of the fit + evaluate hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: fit ONE record
(26 beams x 100 ranges, default order MAXK=4 MAXL=6 -> N=144, curvature regularisation, chi^2 search,
covariance) and evaluate the fitted model on a 128^3 geodetic query grid (BASELINE.json configs[1]).
Inputs (beam geometry, weights/data, query grid, regularisation matrix) are resident in HBM before the
timed region; outputs stay on the device.  With --gpus N > 1 every rank runs the same per-GPU workload on
its own records (independent timesteps: weak scaling, no data-path collective); shared parameters are
broadcast once from rank 0 over RCCL before the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import io
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

CFG = ('[DEFAULT]\nREGULARIZATION_LIST = curvature\nREGULARIZATION_METHOD = chi2\n'
       '[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n')

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
FP64_VALU_PEAK_TF = 78.6       # fp64 vector peak
EVAL_BYTES_PER_POINT = 32.0    # 3 x 8 B coordinates in + 8 B density out
EVAL_FLOPS_PER_POINT = 3.0e3   # SURVEY 8d E1 estimate at the default order


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--grid', type=int, default=128, help='query grid edge (128 -> 128^3 points)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-eval-many', action='store_true', help='skip the secondary many-timesteps evaluation figure')
    ap.add_argument('--no-batched', action='store_true', help='skip the secondary batched-records figure (configs[2])')
    return ap.parse_args()


def cpu_baseline(lat, lon, alt, value, error, R, grid_n):
    """The oracle (faithful CPU restatement of the reference) on a bounded sample of the same workload:
    the full one-record fit, and a 16^3 sub-grid of the query evaluation scaled to grid_n^3 points."""
    import oracle                                  # checker / baseline only - never the measured GPU path
    from volumetricinterp_amd import synth
    import contextlib
    import warnings
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:                              # pragma: no cover
        limiter = contextlib.nullcontext()
    model = oracle.SphHarmLagOracle()
    with limiter, warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t0 = time.perf_counter()
        counter = [0]
        C, dC, c2, params = oracle.fit_records(model, lat, lon, alt, value[:1], error[:1], {'curvature': R},
                                               ['curvature'], counter)
        t_fit = time.perf_counter() - t0
        sub = 16
        g = synth.query_grid(sub)
        Cv = C[0] if np.all(np.isfinite(C[0])) else np.ones(model.nbasis)
        t0 = time.perf_counter()
        oracle.evaluate(model, Cv, *g)
        t_eval_sub = time.perf_counter() - t0
    Q = grid_n**3
    t_eval = t_eval_sub * Q / sub**3
    return dict(value=Q / (t_fit + t_eval), unit='points/s', cores=1, kind='port',
                sample='oracle (NumPy/SciPy restatement, 1 BLAS thread): full fit of 1 record 26x100 N=144 '
                       '(%d eval_C calls, %.1f s) + evaluation of a %d^3 sub-grid (%.2f s) scaled to %d^3 points'
                       % (counter[0] + 1, t_fit, sub, t_eval_sub, grid_n),
                fit_seconds=t_fit, eval_points_per_sec=sub**3 / t_eval_sub)


def main():
    # The contract is ONE JSON line on stdout.  Libraries loaded below (RCCL prints a version banner on
    # communicator creation) write to file descriptor 1 directly, so keep a private handle on the real stdout
    # and point fd 1 at stderr for the rest of the run.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    args = parse_args()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('VINTERP_DEVICE', str(local_rank))

    from volumetricinterp_amd import _lib, synth
    from volumetricinterp_amd.fitengine import FitEngine
    from volumetricinterp_amd.models.sphharmlag import Model

    from volumetricinterp_amd.parallel import Comm
    ctx = _lib.get_context(local_rank)
    # control plane over a Unix socket + ncclBroadcast (RCCL over xGMI) for the shared parameters; no torch in
    # GPU processes (two HIP runtimes in one process crash).  VINTERP_DIST_BACKEND=socket skips RCCL, e.g. to
    # rehearse several ranks on one GPU.
    comm = Comm(backend=os.environ.get('VINTERP_DIST_BACKEND', 'rccl') if world > 1 else None, ctx=ctx)
    model = Model(io.StringIO(CFG), ctx=ctx)
    h = model.handle(ctx)
    N = model.nbasis

    # ---- shared parameters: built on rank 0, broadcast once over RCCL (xGMI) ----------------------------
    nb, nr = synth.GEOM_C2
    P = nb * nr
    shared = {}
    if rank == 0:
        lat, lon, alt = synth.beams(nb, nr, seed=0)
        shared = dict(lat=lat, lon=lon, alt=alt, R=model.eval_reg_matricies['curvature']())
    shared = comm.broadcast_arrays(shared)
    lat, lon, alt, R = shared['lat'], shared['lon'], shared['alt'], shared['R']

    # ---- per-rank inputs, made resident before the timed region -----------------------------------------
    dlat, dlon, dalt = ctx.to_device(lat), ctx.to_device(lon), ctx.to_device(alt)
    At = model.basis_device(dlat, dlon, dalt, P, transposed=True)
    A = At.download().T
    T = 1
    # every rank fits the same synthetic record(s): weak scaling means identical per-GPU work, and the number of
    # Brent iterations (12-40, decided by noise at the default order) differs from record to record
    value, error = synth.synth_records(A, T, seed0=1000)
    W, b = error**-2., value
    npts = [P] * T
    eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
    eng.upload_records(W, b)
    g = synth.query_grid(args.grid)
    Q = g[0].size
    dq = [ctx.to_device(a.ravel()) for a in g]
    dC = ctx.empty((T, N))
    dout = ctx.empty((T, Q))

    fit_ms, eval_ms = [], []

    def step(record=False):
        t0 = time.perf_counter()
        res = eng.fit_resident(npts, calccov=True)
        Cfit = res['Coeffs']
        if not np.all(np.isfinite(Cfit)):          # NaN row (no root): evaluate zeros, work is the same
            Cfit = np.nan_to_num(Cfit)
        dC.upload(Cfit)
        t1 = time.perf_counter()
        ctx.timer_start()
        _lib.check(_lib.lib.vi_eval_f64(h, Q, dq[0].ptr, dq[1].ptr, dq[2].ptr, T, dC.ptr, None, 0, 0., dout.ptr),
                   'vi_eval_f64')
        ctx.timer_stop_ms()
        kms = C.c_double(0.)                       # HIP events on the library's stream, right around the kernel
        _lib.check(_lib.lib.vi_eval_kernel_ms(ctx.handle, C.byref(kms)), 'vi_eval_kernel_ms')
        ems = kms.value
        if record:
            fit_ms.append((t1 - t0) * 1e3)
            eval_ms.append(ems)
        return res

    def barrier():
        ctx.sync()
        comm.barrier()

    for _ in range(args.warmup):
        step()
    ctx.solve_timing(1)            # one HIP event pair around every eigen-solve kernel launch of the timed steps
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(record=True)
    barrier()
    elapsed = comm.max_over_ranks(time.perf_counter() - t0)
    st = ctx.solve_timing(0)

    # ---- secondary figure (not part of `value`): many timesteps on the same grid (SURVEY 8d row E2, configs[3]) ----
    many = None
    if rank == 0 and not args.no_eval_many:
        Tm = 64
        dCm = ctx.to_device(np.random.default_rng(3).standard_normal((Tm, N)))
        dom = ctx.empty((Tm, Q))
        best = float('inf')
        for _ in range(3):
            _lib.check(_lib.lib.vi_eval_f64(h, Q, dq[0].ptr, dq[1].ptr, dq[2].ptr, Tm, dCm.ptr, None, 0, 0., dom.ptr),
                       'vi_eval_f64')
            kms = C.c_double(0.)
            _lib.check(_lib.lib.vi_eval_kernel_ms(ctx.handle, C.byref(kms)), 'vi_eval_kernel_ms')
            best = min(best, kms.value)
        dom.free()
        dCm.free()
        many = {'kernel': 'k_eval_sph_mfma<6,1,2> (v_mfma_f64_16x16x4)', 'timesteps': Tm, 'points': Q, 'ms': best,
                'point_timesteps_per_sec': Q * Tm / (best * 1e-3), 'bound': 'mfma',
                'achieved': 2. * N * Q * Tm / (best * 1e-3) / 1e12, 'peak': FP64_VALU_PEAK_TF, 'unit': 'TFLOP/s',
                'frac': 2. * N * Q * Tm / (best * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                'note': 'algorithmic flops 2N per point-timestep (the contraction only; the recurrence is VALU work on '
                        'top); fp64 MFMA and fp64 VALU share one 78.6 TF peak on gfx950 and do not co-execute'}

    # ---- secondary figure (not part of `value`): configs[2]-style batch - many records of one geometry fitted in one
    #      batch and all evaluated on the same grid (the single-record step above keeps one CU of 256 busy)
    batched = None
    if rank == 0 and not args.no_batched:
        Tb = 256
        vb, eb = synth.synth_records(A, Tb, seed0=5000)
        engb = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
        engb.upload_records(eb**-2., vb)
        dCb = ctx.empty((Tb, N))
        dob = ctx.empty((Tb, Q))
        ctx.sync()
        tb0 = time.perf_counter()
        resb = engb.fit_resident([P] * Tb, calccov=True)
        dCb.upload(np.nan_to_num(resb['Coeffs']))
        tb1 = time.perf_counter()
        _lib.check(_lib.lib.vi_eval_f64(h, Q, dq[0].ptr, dq[1].ptr, dq[2].ptr, Tb, dCb.ptr, None, 0, 0., dob.ptr),
                   'vi_eval_f64')
        ctx.sync()
        tb2 = time.perf_counter()
        ocb = resb['search']['curvature']['outcomes']
        batched = {'records': Tb, 'fit_ms': (tb1 - tb0) * 1e3, 'eval_ms': (tb2 - tb1) * 1e3,
                   'records_per_sec': Tb / (tb2 - tb0), 'points_per_sec': Tb * Q / (tb2 - tb0),
                   'outcomes': {o_: ocb.count(o_) for o_ in set(ocb)},
                   'note': 'single pass without warm-up: %d records of the bench geometry fitted as one batch (chi2 search, '
                           'covariance) and each evaluated on the %d^3 grid' % (Tb, args.grid)}
        dob.free()
        dCb.free()
        engb.close()

    if rank == 0:
        ev = float(np.mean(eval_ms)) if eval_ms else float('nan')
        traffic = None                 # HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/
        try:
            pmc = json.load(open(os.path.join(REPO, 'profiles', 'r1_eval_pmc.json')))
            if pmc.get('points_per_launch') == Q * T:
                traffic = pmc['hbm_bytes_per_launch']
        except Exception:
            pass
        pts = args.steps * Q * T * world
        out = {
            'metric': 'fit+eval query-points/sec', 'value': pts / elapsed, 'unit': 'points/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: per GPU 1 record, 26-beam x 100-range fit (N=144, curvature, chi2 '
                                   'search, covariance) + %d^3 geodetic query grid, fp64' % args.grid,
                       'points_per_step_per_gpu': Q * T, 'timesteps_per_step_per_gpu': T},
            'timesteps_per_sec': args.steps * T * world / elapsed,
            'breakdown_ms': {'fit': float(np.mean(fit_ms)), 'eval_kernel': ev,
                             'fit_solves_per_step': eng.stats['solves'] / max(1, args.steps + args.warmup),
                             'fit_outcome': res['search']['curvature']['outcomes'],
                             'root_finder': [i.get('finder') for i in res['search']['curvature']['info']]},
            'eval_points_per_sec_per_gpu': Q * T / (ev * 1e-3),
            'comm': {'backend': comm.backend, 'rccl_broadcast': bool(comm.rccl_ready), 'notes': comm.notes},
            'roofline': {'kernel': 'k_eval_sph_fast<6,4,1>', 'bound': 'hbm',
                         'achieved': EVAL_BYTES_PER_POINT * Q * T / (ev * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': EVAL_BYTES_PER_POINT * Q * T / (ev * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         'traffic': traffic,
                         'note': 'fused eval is fp64-VALU-bound (AI ~94 flop/B): %.2f of the %.1f TF fp64 vector peak '
                                 'at ~3.0 kflop/point' % (EVAL_FLOPS_PER_POINT * Q * T / (ev * 1e-3) / 1e12
                                                          / FP64_VALU_PEAK_TF, FP64_VALU_PEAK_TF)},
        }
        # the kernel the step actually spends its time in (latency-bound small-matrix
        # factorisations, neither HBM nor MFMA): one 512-thread workgroup = one CU per system, so a single-record
        # step can occupy 1 of 256 CUs during the ~15 dependent root-finder solves
        if st['timed']:
            flops = 10. * N**3 * st['systems'] * st['timed'] / max(1, st['launches'])
            out['fit_kernel'] = {
                'kernel': 'k_jacobi_solve<5>', 'launches_per_step': st['launches'] / args.steps,
                'systems_per_step': st['systems'] / args.steps, 'avg_launch_ms': st['total_ms'] / st['timed'],
                'max_launch_ms': st['max_ms'], 'ms_per_step': st['total_ms'] / st['timed'] * st['launches'] / args.steps,
                'share_of_step': st['total_ms'] / st['timed'] * st['launches'] / (elapsed * 1e3),
                'bound': 'LDS-resident eigen-solve, one CU per system (latency-bound at 1 record)',
                'achieved_gflops': flops / (st['total_ms'] * 1e-3) / 1e9, 'flops_model': '10 N^3 per solve (SURVEY 8d F2)',
                'peak_gflops': FP64_VALU_PEAK_TF * 1e3,
                'frac': flops / (st['total_ms'] * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                'frac_note': 'of the whole-chip fp64 peak; a launch of B systems can occupy min(B, 256) of 256 CUs'}
        if many is not None:
            out['eval_many_timesteps'] = many
        if batched is not None:
            out['batched_records'] = batched
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(lat, lon, alt, value, error, R, args.grid)
        elif world == 1:
            out['cpu_baseline'] = None
        real_stdout.write(json.dumps(out) + '\n')
        real_stdout.flush()
    comm.close()
    if any('did not return' in n for n in comm.notes):
        # a communicator bootstrap that never returned leaves a thread inside RCCL: skip the interpreter's and the
        # runtime's finalisers, which could block on it, now that the result line is out
        sys.stderr.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
