#!/usr/bin/env python3
"""Benchmark of the fit + evaluate hot path on MI355X.

ONE workload at every N (round 4; VERDICT round 3: the N = 1 and the N > 1 lines used to be different workloads, so no scaling
curve could be formed): **workload c3 = BASELINE.json configs[3]** - 10000 timesteps (26 beams x 100 ranges, default order
MAXK=4 MAXL=6 -> N=144, curvature regularisation, chi^2 search, covariance) sharded ceil(T/N) per rank - independent records, no
data-path collective -, each rank fits its shard as one batch and evaluates EVERY timestep of it on a 256^3 geodetic grid with
the hull mask (basis matrix of the grid resident in HBM, matrix-core product K2r over 512 timesteps per call), all of it inside
the timed region.  A "step" is one such pass; strong scaling; value = timesteps/s = T / (barrier-to-barrier wall time per
step, max over ranks).  The line carries
  roofline       the dominant part of the step, the FIT: the in-LDS eigen-solves K3 (k_jacobi_solve launches + the Jacobi rounds
                 inside k_brent_warm), algorithmic LDS bytes counted by the kernels themselves over HIP-event launch durations;
  roofline_eval  the evaluation product K2r against the fp64 matrix peak;
  cpu_baseline   (N = 1) the oracle - the faithful CPU restatement of the reference - on 16 records + a hull-masked sub-grid,
                 extrapolated to timesteps/s;
and, at N = 1, secondary objects that are NOT part of `value`: `single_record` (BASELINE configs[1], the latency path: one
record + 128^3 grid per step, mean / median / max over sixteen distinct records, with the K3 launch figures of that path),
`roofline_eval_fused` (the fused evaluation kernel with and without the hull pass), `eval_many_timesteps`, `batched_records`
(configs[2]: 1000 records in one batch).
`--workload c1` prints the configs[1] line on its own (weak-scaling replica per rank), as rounds 1-3 did at N = 1.
Inputs (beam geometry, weights/data, query grid, regularisation matrix, hull facets) are resident in HBM before the
timed region; outputs stay on the device.  Shared parameters are broadcast once from rank 0 over RCCL before the timed
region; if RCCL was to be used and did not come up, the line says so and the exit status is 3.

`python bench.py --gpus N` without a torchrun environment starts N child processes itself (one per device, before
anything touches a GPU) and prints the line rank 0 produced.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import io
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

CFG = ('[DEFAULT]\nREGULARIZATION_LIST = curvature\nREGULARIZATION_METHOD = chi2\n'
       '[MODEL]\nNAME = sphharmlag\nMAXK = 4\nMAXL = 6\nCAP_LIM = 10\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n')

# ---- roofline constants (MI355X_MICROARCH.md) --------------------------------------------------------------
HBM_PEAK_GBS = 8000.0          # HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
FP64_PEAK_TF = 78.6            # fp64 vector = fp64 matrix peak (one set of DP units)
N_CU = 256
CLOCK_GHZ = 2.4
# LDS, per CU: ds_read_b64 256 B/clk, ds_write_b64 ~85 B/clk.  A Jacobi round moves every byte of the matrix once out
# of and once back into LDS, so the combined rate for equal read and write volumes is 2 / (1/256 + 1/85) B/clk.
LDS_RW_BYTES_PER_CLK_CU = 2. / (1. / 256. + 1. / 85.)
LDS_PEAK_GBS = LDS_RW_BYTES_PER_CLK_CU * CLOCK_GHZ * N_CU          # ~78 TB/s, all 256 CUs streaming
EVAL_BYTES_PER_POINT = 32.0    # 3 x 8 B coordinates in + 8 B density out
EVAL_FLOPS_PER_POINT = 3.0e3   # SURVEY 8d E1 estimate at the default order
# K2r, one call of 256 timesteps on 256^3 points: HBM bytes read + written from the PMC passes of profiles/r3_k2r_pmc.txt
K2R_PMC_BYTES_PER_256_TIMESTEPS = 30.2e9 + 34.4e9
K2R_PMC_BYTES_PER_512_TIMESTEPS = 34.55e9 + 68.74e9          # same passes (profiles/r3_k2r_pmc.txt), 512 timesteps per call


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='default: 2 (workload c3), 16 (workload c1)')
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', choices=('c1', 'c3'), default='c3',
                    help="c3 (default at every N): BASELINE configs[3], 10000 timesteps sharded over the ranks, 256^3 grid; "
                         "c1: configs[1] on its own, one record + 128^3 grid per step and GPU (weak-scaling replica)")
    ap.add_argument('--records', type=int, default=10000, help='timesteps of workload c3 (all ranks together)')
    ap.add_argument('--grid', type=int, default=128, help='query grid edge of the configs[1] figures (128 -> 128^3 points)')
    ap.add_argument('--c3-grid', type=int, default=256, help='query grid edge of workload c3 (rehearsals: smaller)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='workload c3 at N = 1: skip the secondary objects')
    ap.add_argument('--no-eval-many', action='store_true', help='skip the secondary many-timesteps evaluation figure')
    ap.add_argument('--no-batched', action='store_true', help='skip the secondary batched-records figure (configs[2])')
    ap.add_argument('--batched-records', type=int, default=1000, help='records of the configs[2] figure')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 16 if args.workload == 'c1' else 2
    return args


# ---- --gpus N without a launcher: start the ranks ourselves ---------------------------------------------------
def spawn_ranks(n):
    """Start n fresh copies of this script, one per device (RANK = LOCAL_RANK = i), wait for them and relay rank 0's
    line.  Runs before anything in this process has touched a GPU (no HIP call, no library load): children are
    ordinary subprocesses, nothing is exec'ed over an initialised runtime."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


# ---- CPU baseline (rank 0, N = 1 only) -------------------------------------------------------------------------
def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _cpu_worker(job):
    """One worker of the all-cores run: the faithful oracle fit of one record + a hull-masked evaluation of a
    sub-grid.  Runs in a fresh process (spawn) with one BLAS thread."""
    seed, sub, variant = job
    import warnings
    import oracle                                  # checker / baseline only - never the measured GPU path
    from oracle import fit_fast
    from volumetricinterp_amd import synth
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:                              # pragma: no cover
        pass
    model = oracle.SphHarmLagOracle()
    lat, lon, alt = synth.beams(*synth.GEOM_C2, seed=0)
    R = np.load(os.path.join(REPO, 'tests', 'golden', 'regmat.npz'))['default_curvature']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        A = model.basis(lat, lon, alt)
        b, err, _ = synth.synth_record(A, seed)
        counter = [0]
        t0 = time.perf_counter()
        if variant == 'faithful':
            Cf, _, _, _ = oracle.fit_records(model, lat, lon, alt, b[None], err[None], {'curvature': R}, ['curvature'],
                                             counter)
            Cv = Cf[0]
        else:
            Cv, _, _, _ = fit_fast.fit_record(A, b, err**-2., R, counter)
        t_fit = time.perf_counter() - t0
        g = synth.query_grid(sub)
        hv = oracle.compute_hull_vertices(lat, lon, alt)
        Cv = Cv if np.all(np.isfinite(Cv)) else np.ones(model.nbasis)
        t0 = time.perf_counter()
        oracle.evaluate(model, Cv, *g, hull_vert=hv)       # check_hull=True, one Qhull per point (estimate.py:153-178)
        t_eval = time.perf_counter() - t0
        t0 = time.perf_counter()
        oracle.evaluate(model, Cv, *g)
        t_eval_nohull = time.perf_counter() - t0
    return dict(t_fit=t_fit, t_eval=t_eval, t_eval_nohull=t_eval_nohull, eval_C_calls=counter[0] + 1)


def cpu_baseline(grid_n, unit='points/s'):
    """The oracle (faithful CPU restatement of the reference, kind 'port') on a bounded sample of the same workload,
    as BASELINE.md section 4 asks: one process, then `multiprocessing` over independent records on the box's host-core share;
    plus the optimised-CPU variant for context.  Sample per worker: the full one-record fit (26 x 100, N = 144) and a
    hull-masked evaluation of a 12^3 sub-grid scaled to grid_n^3 points.  unit 'timesteps/s' (workload c3): a timestep = that
    fit + that evaluation, records being independent the rate of W workers is the sum of their own rates; 'points/s'
    (workload c1): the same time per timestep, counted in evaluated points."""
    import multiprocessing as mp
    sub = 12
    Q = grid_n**3
    ncpu = os.cpu_count() or 1
    try:
        ncpu_avail = len(os.sched_getaffinity(0))
    except AttributeError:                          # pragma: no cover
        ncpu_avail = ncpu
    # the GPU box gives a one-GPU job a 16-core share of the host whatever the affinity mask says
    workers = max(1, int(os.environ.get("VINTERP_CPU_WORKERS", min(ncpu_avail, 16))))
    ctxmp = mp.get_context('spawn')                 # fresh interpreters: nothing of the GPU runtime is inherited

    def rate(r):
        return (Q if unit == 'points/s' else 1.) / (r['t_fit'] + r['t_eval'] * Q / sub**3)
    # (map_async + a deadline: a worker that dies - it happened once under a profiler's preloaded tool - would leave
    # pool.map waiting for ever; the bench then reports the baseline as missing instead of hanging)
    with ctxmp.Pool(1) as pool:
        one = pool.map_async(_cpu_worker, [(1000, sub, 'faithful')]).get(timeout=240)[0]
        fast = pool.map_async(_cpu_worker, [(1000, sub, 'optimised')]).get(timeout=120)[0]
    t0 = time.perf_counter()
    with ctxmp.Pool(workers) as pool:
        many = pool.map_async(_cpu_worker, [(1000 + i, sub, 'faithful') for i in range(workers)], chunksize=1).get(timeout=480)
    wall = time.perf_counter() - t0
    all_rate = sum(rate(r) for r in many)           # every worker's own step rate under full load
    return dict(value=all_rate, unit=unit, cores=workers, kind='port',
                sample='oracle (NumPy/SciPy restatement of the reference, one BLAS thread per process): %d processes = %d '
                       'records, each the full fit of one 26x100 record at N=144 (%d eval_C calls, %.1f s alone, %.1f s mean '
                       'under load) + a hull-masked (one Qhull per point) evaluation of a %d^3 sub-grid (%.2f s) scaled to '
                       '%d^3 points (%.0f s per timestep); value = sum over the processes of 1 / (their own time per timestep)'
                       '%s; wall %.1f s'
                       % (workers, workers, one['eval_C_calls'], one['t_fit'], float(np.mean([r['t_fit'] for r in many])), sub,
                          one['t_eval'], grid_n, float(np.mean([r['t_fit'] + r['t_eval'] * Q / sub**3 for r in many])),
                          ' x points per timestep' if unit == 'points/s' else '', wall),
                host_cpu_count=ncpu, host_cpus_available=ncpu_avail, cpu_model=_cpu_model(), blas_threads_per_process=1,
                single_process={'value': rate(one), 'cores': 1, 'fit_seconds': one['t_fit'],
                                'eval_points_per_sec': sub**3 / one['t_eval'],
                                'eval_points_per_sec_no_hull': sub**3 / one['t_eval_nohull'],
                                'timesteps_per_sec': 1. / one['t_fit']},
                optimised_single_process={'value': rate(fast), 'cores': 1, 'fit_seconds': fast['t_fit'],
                                          'eval_C_calls': fast['eval_C_calls'],
                                          'note': 'A^T W A once per record, chi^2(alpha) memoised across scale factors; '
                                                  'same LAPACK calls on the same matrices (context only)'},
                core_share={'processes': workers, 'fit_timesteps_per_sec': workers / float(np.mean([r['t_fit'] for r in many])),
                            'fit_seconds_mean': float(np.mean([r['t_fit'] for r in many])),
                            'note': 'fit only (no evaluation), all processes together; the processes are the 16-core share '
                                    'a one-GPU job gets of the host, not all %d hardware threads' % ncpu})


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    sys.exit(run(args))


def run(args):
    # The contract is ONE JSON line on stdout.  Libraries loaded below (RCCL prints a version banner on
    # communicator creation) write to file descriptor 1 directly, so keep a private handle on the real stdout
    # and point fd 1 at stderr for the rest of the run.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('VINTERP_DEVICE', str(local_rank))
    # CPU baseline first (rank 0, N = 1 only): its worker processes are started before this process has loaded the
    # HIP runtime, and nothing runs on the GPU while the host cores are being timed
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args.grid, 'points/s') if args.workload == 'c1' else cpu_baseline(args.c3_grid, 'timesteps/s')
        except Exception as e:                       # a timed-out or dead worker: say so in the line, do not hang or die
            cpu = dict(value=None, unit='points/s' if args.workload == 'c1' else 'timesteps/s', cores=0, kind='port',
                       sample='not measured: %s: %s' % (type(e).__name__, e))

    from volumetricinterp_amd import _lib, synth
    from volumetricinterp_amd.estimate import hull_equations
    from volumetricinterp_amd.fitengine import FitEngine
    from volumetricinterp_amd.geodesy import geodetic2ecef
    from volumetricinterp_amd.models.sphharmlag import Model
    from volumetricinterp_amd.parallel import Comm
    from scipy.spatial import ConvexHull

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
        Rpts = np.array(geodetic2ecef(lat, lon, alt)).T
        eq, tol = hull_equations(Rpts[ConvexHull(Rpts).vertices])          # interpolate.py:409-426 + estimate.py:153-178
        shared = dict(lat=lat, lon=lon, alt=alt, R=model.eval_reg_matricies['curvature'](), hull_eq=eq,
                      hull_tol=np.array([tol]))
    shared = comm.broadcast_arrays(shared)
    lat, lon, alt, R = shared['lat'], shared['lon'], shared['alt'], shared['R']
    hull_eq, hull_tol = np.ascontiguousarray(shared['hull_eq']), float(shared['hull_tol'][0])
    F = hull_eq.shape[0]

    # ---- per-rank inputs, made resident before the timed region -----------------------------------------
    dlat, dlon, dalt = ctx.to_device(lat), ctx.to_device(lon), ctx.to_device(alt)
    At = model.basis_device(dlat, dlon, dalt, P, transposed=True)
    A = At.download().T
    if args.workload == 'c3':
        out = run_c3(args, ctx, comm, model, h, At, A, R, hull_eq, hull_tol, rank, world)
        if rank == 0 and world == 1:
            out['cpu_baseline'] = cpu
            if not args.no_secondary:
                # secondary objects (not part of `value`): the single-record latency path of configs[1] and the evaluation
                # kernels on their own, measured after the timed region of workload c3
                c1 = run_c1(args, ctx, comm, model, h, At, A, R, hull_eq, hull_tol, rank, world, steps=16, warmup=1)
                out['single_record'] = {
                    'workload': c1['config']['workload'], 'points_per_sec': c1['value'], 'ms_per_step': c1['ms_per_step'],
                    'steps': c1['steps'], 'step_ms': c1['step_ms'], 'guard_verdicts': c1['guard_verdicts'],
                    'breakdown_ms': c1['breakdown_ms'], 'steps_detail': c1['steps_detail'], 'roofline': c1.get('roofline')}
                out['roofline_eval_fused'] = c1['roofline_eval']
                for k in ('eval_many_timesteps', 'batched_records'):
                    if k in c1:
                        out[k] = c1[k]
        if rank == 0:
            real_stdout.write(json.dumps(out) + '\n')
            real_stdout.flush()
        return finish(comm, world)
    out = run_c1(args, ctx, comm, model, h, At, A, R, hull_eq, hull_tol, rank, world, steps=args.steps, warmup=args.warmup)
    if rank == 0:
        if world == 1:
            out['cpu_baseline'] = cpu
        real_stdout.write(json.dumps(out) + '\n')
        real_stdout.flush()
    return finish(comm, world)


def run_c1(args, ctx, comm, model, h, At, A, R, hull_eq, hull_tol, rank, world, steps, warmup):
    """BASELINE configs[1], the single-record latency path: a step = fit ONE record (26 x 100, N = 144, curvature, chi^2
    search, covariance) + evaluate it on a 128^3 grid with the hull mask; step t fits record t mod 16 of sixteen distinct
    resident records.  Returns the line of `--workload c1` (rank 0; None elsewhere); workload c3 at N = 1 embeds parts of it
    as secondary objects."""
    from volumetricinterp_amd import _lib, synth
    from volumetricinterp_amd.fitengine import FitEngine
    N, P = model.nbasis, A.shape[0]
    F = hull_eq.shape[0]
    T = 1
    # Sixteen distinct synthetic records, each resident in its own engine (one record per step: the single-record latency
    # path); the engines share their scratch buffers.  Every rank fits the same records: weak scaling means identical
    # per-GPU work.  The number of root-finder iterations differs from record to record (12-60: where chi^2(alpha) jumps
    # Brent bisects down to its 2e-12), so one hand-picked record says little.
    NREC = 16
    value, error = synth.synth_records(A, NREC, seed0=1000)
    npts = [P] * T
    engs = []
    for r in range(NREC):
        e_ = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'], scratch_of=engs[0] if engs else None)
        e_.upload_records(error[r:r + 1]**-2., value[r:r + 1])
        engs.append(e_)
    eng = engs[0]
    g = synth.query_grid(args.grid)
    Q = g[0].size
    dq = [ctx.to_device(a.ravel()) for a in g]
    dhull = ctx.to_device(hull_eq)
    dC = ctx.empty((T, N))
    dout = ctx.empty((T, Q))

    fit_ms, eval_ms, step_ms, verdicts, step_info = [], [], [], [], []

    def eval_grid(dCx, Tx, doutx, hull=True):
        _lib.check(_lib.lib.vi_eval_f64(h, Q, dq[0].ptr, dq[1].ptr, dq[2].ptr, Tx, dCx.ptr,
                                        dhull.ptr if hull else None, F if hull else 0, hull_tol if hull else 0.,
                                        doutx.ptr), 'vi_eval_f64')

    def verdict(res):
        """What the engine's consistency guard said about the record."""
        oc = res['search']['curvature']['outcomes'][0]
        if oc != 'root':
            return oc, None
        i0 = res['search']['curvature']['info'][0]
        if i0.get('consistent'):
            v = 'consistent'
        elif i0.get('jump'):
            v = 'jump'
        elif i0.get('redone_cold'):
            v = 'redone_cold'
        elif 0 in res['search']['curvature'].get('polished_cold', []):
            v = 'polished_cold'
        else:
            v = 'band'                           # 1e-6 nu < |chi^2 - nu| <= 1e-4 nu: the noise of the similarity transform
        return v, i0

    def step(i, record=False):
        e_ = engs[i % NREC]
        t0 = time.perf_counter()
        res = e_.fit_resident(npts, calccov=True)
        Cfit = res['Coeffs']
        if not np.all(np.isfinite(Cfit)):          # NaN row (no root): evaluate zeros, work is the same
            Cfit = np.nan_to_num(Cfit)
        dC.upload(Cfit)
        t1 = time.perf_counter()
        ctx.timer_start()                          # HIP events on the library's own stream, around the whole call
        eval_grid(dC, T, dout, hull=True)          # hull mask on: the reference default (estimate.py:75)
        ems = ctx.timer_stop_ms()
        t2 = time.perf_counter()
        if record:
            fit_ms.append((t1 - t0) * 1e3)
            eval_ms.append(ems)
            step_ms.append((t2 - t0) * 1e3)
            v, i0 = verdict(res)
            verdicts.append(v)
            step_info.append({'record': i % NREC, 'ms': (t2 - t0) * 1e3, 'verdict': v,
                              'iterations': i0.get('iterations') if i0 else None,
                              'chi2_minus_nu': i0.get('chi2_minus_nu') if i0 else None,
                              'log10_alpha': (float(np.log10(res['reg_params'][0]['curvature']))
                                              if v not in ('no_root', 'too_smooth', 'skipped') and
                                              res['reg_params'][0]['curvature'] > 0 else None)})
        return res

    def barrier():
        ctx.sync()
        comm.barrier()

    for i in range(warmup):
        step(i)
    ctx.solve_timing(1)            # one HIP event pair around every eigen-solve kernel launch of the timed steps
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        res = step(i, record=True)
    barrier()
    elapsed = comm.max_over_ranks(time.perf_counter() - t0)
    st = ctx.solve_timing(0)
    solves_per_step = sum(e_.stats['solves'] for e_ in engs) / max(1, steps + warmup)

    def kernel_ms(fn, reps=5):
        best = float('inf')
        ctx.eval_timing(True)                      # HIP events right around the evaluation kernel launches: off by default
        try:
            for _ in range(reps):
                fn()
                best = min(best, ctx.eval_kernel_ms())
        finally:
            ctx.eval_timing(False)
        return best

    # ---- the evaluation kernel on its own (not part of `value`): with and without the hull pass ---------------------
    ev_kernel_hull = ev_kernel_nohull = ev_call_nohull = float('nan')
    if rank == 0:
        ev_kernel_hull = kernel_ms(lambda: eval_grid(dC, T, dout, hull=True))
        ev_kernel_nohull = kernel_ms(lambda: eval_grid(dC, T, dout, hull=False))
        best = float('inf')
        for _ in range(5):
            ctx.timer_start()
            eval_grid(dC, T, dout, hull=False)
            best = min(best, ctx.timer_stop_ms())
        ev_call_nohull = best

    # ---- secondary figure (not part of `value`): many timesteps on the same grid (SURVEY 8d row E2, configs[3]) ----
    many = None
    if rank == 0 and not args.no_eval_many:
        Tm = 64
        dCm = ctx.to_device(np.random.default_rng(3).standard_normal((Tm, N)))
        dom = ctx.empty((Tm, Q))
        best = kernel_ms(lambda: eval_grid(dCm, Tm, dom, hull=True), reps=3)
        dom.free()
        dCm.free()
        many = {'kernel': 'k_eval_sph_mfma<6,1,2> (v_mfma_f64_16x16x4)', 'timesteps': Tm, 'points': Q, 'ms': best,
                'hull_mask': True, 'point_timesteps_per_sec': Q * Tm / (best * 1e-3), 'bound': 'mfma',
                'achieved': 2. * N * Q * Tm / (best * 1e-3) / 1e12, 'peak': FP64_PEAK_TF, 'unit': 'TFLOP/s',
                'frac': 2. * N * Q * Tm / (best * 1e-3) / 1e12 / FP64_PEAK_TF,
                'note': 'algorithmic flops 2N per point-timestep (the contraction only; the recurrence is VALU work on '
                        'top); fp64 MFMA and fp64 VALU share one 78.6 TF peak on gfx950 and do not co-execute'}

        # the same many-timesteps evaluation from the RESIDENT basis matrix of the grid (K2r, csrc/vi_eval_resident.hip; what
        # workload c3 runs): basis once (hull mask as NaN rows), then one matrix-core product per call of 256 timesteps
        Tr = 256
        dY = ctx.empty((N, Q))
        dCr = ctx.to_device(np.random.default_rng(4).standard_normal((Tr, N)))
        dor = ctx.empty((Tr, Q))
        ctx.timer_start()
        _lib.check(_lib.lib.vi_eval_basis_f64(h, Q, dq[0].ptr, dq[1].ptr, dq[2].ptr, dhull.ptr, F, hull_tol, dY.ptr),
                   'vi_eval_basis_f64')
        basis_ms = ctx.timer_stop_ms()
        bestr = float('inf')
        for _ in range(3):
            ctx.timer_start()
            _lib.check(_lib.lib.vi_eval_resident_f64(h, Q, Tr, dY.ptr, dCr.ptr, dor.ptr), 'vi_eval_resident_f64')
            bestr = min(bestr, ctx.timer_stop_ms())
        for b_ in (dY, dCr, dor):
            b_.free()
        many['resident_basis'] = {
            'kernel': 'k_eval_resident (K2r: v_mfma_f64_16x16x4 on the basis matrix of the grid kept in HBM)', 'timesteps': Tr,
            'points': Q, 'basis_ms_once': basis_ms, 'basis_bytes': int(N) * int(Q) * 8, 'ms': bestr, 'hull_mask': True,
            'point_timesteps_per_sec': Q * Tr / (bestr * 1e-3), 'bound': 'mfma',
            'achieved': 2. * N * Q * Tr / (bestr * 1e-3) / 1e12, 'peak': FP64_PEAK_TF, 'unit': 'TFLOP/s',
            'frac': 2. * N * Q * Tr / (bestr * 1e-3) / 1e12 / FP64_PEAK_TF,
            'note': 'HIP events around the call on the library\'s stream; on the 256^3 grid of workload c3 (where the 73 KB '
                    'coefficient tile of a workgroup is reused for 32 x 256 points instead of 4 x 256) 60-62 TF'}

    # ---- secondary figure (not part of `value`): BASELINE configs[2] - 1000 records of one geometry fitted as one
    #      batch (the single-record step above keeps one CU of 256 busy).  One warm-up pass, one timed pass.
    batched = None
    if rank == 0 and not args.no_batched and args.batched_records > 0:
        Tb = args.batched_records
        vb, eb = synth.synth_records(A, Tb, seed0=5000)
        engb = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
        engb.upload_records(eb**-2., vb)
        engb.fit_resident([P] * Tb, calccov=True)                    # warm-up (allocations, rocBLAS kernels)
        engb.solve_timing(1)
        ctx.sync()
        tb0 = time.perf_counter()
        resb = engb.fit_resident([P] * Tb, calccov=True)
        ctx.sync()
        tb1 = time.perf_counter()
        stb = engb.solve_timing(0)
        ocb = resb['search']['curvature']['outcomes']
        Te = min(Tb, 64)                                             # evaluation of a tile of the fitted records
        dCb = ctx.to_device(np.nan_to_num(resb['Coeffs'][:Te]))
        dob = ctx.empty((Te, Q))
        evb = kernel_ms(lambda: eval_grid(dCb, Te, dob, hull=True), reps=2) * Tb / Te
        lds_b = 2. * 8. * (N * (N + 1) // 2) * stb['rounds']
        npipe = engb.stats.get('pipelines', 1)
        k3_ms = stb['total_ms'] * stb['launches'] / max(1, stb['timed'])     # summed over the pipelines (they overlap)
        busy_ms = min(k3_ms, (tb1 - tb0) * 1e3)
        batched = {'records': Tb, 'fit_ms': (tb1 - tb0) * 1e3, 'records_per_sec_fit': Tb / (tb1 - tb0),
                   'eval_ms_scaled': evb, 'records_per_sec': Tb / (tb1 - tb0 + evb * 1e-3),
                   'points_per_sec': Tb * Q / (tb1 - tb0 + evb * 1e-3),
                   'pipelines': npipe, 'solves': stb['systems'], 'solve_launches': stb['launches'],
                   'jacobi_kernel_ms_summed': k3_ms,
                   'jacobi_lds_gbs': lds_b / max(1e-9, busy_ms * 1e-3) / 1e9,
                   'jacobi_lds_frac': lds_b / max(1e-9, busy_ms * 1e-3) / 1e9 / LDS_PEAK_GBS,
                   'jacobi_lds_note': 'algorithmic LDS bytes of all K3 launches - k_jacobi_solve and the Jacobi rounds inside '
                                      'k_brent_warm (one launch per pipeline: Brent\'s iteration of its records, whose whole '
                                      'duration is counted, chi^2 and re-basing included) - over the time they were running: '
                                      'the sum of the launch durations, capped at the wall time of the fit when the launches '
                                      'of concurrent pipelines overlap; solves = systems of the k_jacobi_solve launches + '
                                      'records of the k_brent_warm launches',
                   'outcomes': {o_: ocb.count(o_) for o_ in set(ocb)},
                   'redone_cold': len(resb['search']['curvature'].get('redone_cold', [])),
                   'note': 'configs[2]: %d records of the bench geometry fitted as one batch (chi2 search, covariance), '
                           'timed pass after one warm-up pass; evaluation of %d of them on the %d^3 grid (hull mask on), '
                           'scaled to all' % (Tb, Te, args.grid)}
        dob.free()
        dCb.free()
        engb.close()

    if rank == 0:
        ev = float(np.mean(eval_ms)) if eval_ms else float('nan')
        traffic_eval = None            # HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/
        try:
            pmc = json.load(open(os.path.join(REPO, 'profiles', 'r1_eval_pmc.json')))
            if pmc.get('points_per_launch') == Q * T:
                traffic_eval = pmc['hbm_bytes_per_launch']
        except Exception:
            pass
        pts = steps * Q * T * world
        out = {
            'metric': 'fit+eval query-points/sec', 'value': pts / elapsed, 'unit': 'points/s',
            'n_gpus': world, 'steps': steps, 'warmup': warmup,
            'ms_per_step': elapsed / steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: per GPU and step 1 record (step t fits record t mod %d of %d distinct resident '
                                   'records), 26-beam x 100-range fit (N=144, curvature, chi2 search, covariance) + %d^3 '
                                   'geodetic query grid with the hull mask, fp64' % (NREC, NREC, args.grid),
                       'points_per_step_per_gpu': Q * T, 'timesteps_per_step_per_gpu': T, 'distinct_records': NREC},
            'timesteps_per_sec': steps * T * world / elapsed,
            'step_ms': {'mean': float(np.mean(step_ms)), 'median': float(np.median(step_ms)), 'min': float(np.min(step_ms)),
                        'max': float(np.max(step_ms)), 'note': 'rank 0, per step (host clock around fit + evaluation); '
                                                               'ms_per_step is the mean over the timed steps'},
            'guard_verdicts': {v: verdicts.count(v) for v in sorted(set(verdicts))},
            'steps_detail': step_info,
            'breakdown_ms': {'fit': float(np.mean(fit_ms)), 'eval_call_hull_on': ev,
                             'fit_solves_per_step': solves_per_step,
                             'brent_iterations_mean': float(np.mean([d['iterations'] for d in step_info
                                                                     if d['iterations'] is not None] or [float('nan')]))},
            'eval_points_per_sec_per_gpu': Q * T / (ev * 1e-3),
            'comm': {'backend': comm.backend, 'rccl_broadcast': bool(comm.rccl_ready), 'notes': comm.notes},
        }
        # ---- roofline of the kernel that dominates the step: the in-LDS Jacobi eigen-solve (K3).  Neither HBM nor
        #      MFMA bound it: the matrix lives in one CU's LDS and every round moves it once through the registers.
        #      Algorithmic bytes: 2 (read + write) x 8 B x N(N+1)/2 per round and system; rounds are
        #      counted by the kernel itself; durations are HIP events around every launch of the timed steps.
        if st['timed']:
            avg_ms = st['total_ms'] / st['timed']
            launches_per_step = st['launches'] / steps
            lds_bytes_per_launch = 2. * 8. * (N * (N + 1) // 2) * st['rounds'] / max(1, st['launches'])
            flops_per_launch = 10. * N**3 * st['systems'] / max(1, st['launches'])
            ach = lds_bytes_per_launch / (avg_ms * 1e-3) / 1e9
            sys_per_launch = st['systems'] / max(1, st['launches'])
            out['roofline'] = {
                'kernel': 'k_jacobi_solve', 'bound': 'lds', 'achieved': ach, 'peak': LDS_PEAK_GBS, 'unit': 'GB/s',
                'frac': ach / LDS_PEAK_GBS, 'traffic': None,
                'traffic_note': 'HBM traffic is not the resource: per system 166 KB in, ~1.3 MB of rotation log out and '
                                'back, against ~0.1-0.5 GB through LDS',
                'launches_per_step': launches_per_step, 'systems_per_launch': sys_per_launch,
                'avg_launch_ms': avg_ms, 'max_launch_ms': st['max_ms'], 'ms_per_step': avg_ms * launches_per_step,
                'share_of_step': avg_ms * launches_per_step / (elapsed / steps * 1e3),
                'rounds_per_system': st['rounds'] / max(1, st['systems']),
                'peak_note': 'chip-wide LDS rate for equal read and write volumes: %.0f B/clk/CU x %.1f GHz x %d CUs; one '
                             'system occupies one CU, so a launch of B systems can reach at most min(B,256)/256 of it'
                             % (LDS_RW_BYTES_PER_CLK_CU, CLOCK_GHZ, N_CU),
                'frac_of_occupied_cus': ach / (LDS_PEAK_GBS * min(1., sys_per_launch / N_CU)),
                'fp64': {'achieved_tflops': flops_per_launch / (avg_ms * 1e-3) / 1e12, 'peak_tflops': FP64_PEAK_TF,
                         'frac': flops_per_launch / (avg_ms * 1e-3) / 1e12 / FP64_PEAK_TF,
                         'flops_model': '10 N^3 per solve (SURVEY 8d F2)'}}
        # ---- second object: the fused evaluation kernel (0.5 % of the step, the headline kernel of the evaluate half)
        out['roofline_eval'] = {
            'kernel': 'k_eval_sph_fast<6,4,1> (+ k_hull_mask_mx)', 'bound': 'fp64 VALU (AI ~94 flop/B >> 9.8)',
            'kernel_ms_hull_on': ev_kernel_hull, 'kernel_ms_hull_off': ev_kernel_nohull, 'call_ms_hull_off': ev_call_nohull,
            'call_ms_hull_on': ev,
            'hbm': {'achieved': EVAL_BYTES_PER_POINT * Q * T / (ev_kernel_nohull * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': EVAL_BYTES_PER_POINT * Q * T / (ev_kernel_nohull * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    'traffic': traffic_eval},
            'valu': {'achieved': EVAL_FLOPS_PER_POINT * Q * T / (ev_kernel_nohull * 1e-3) / 1e12, 'peak': FP64_PEAK_TF,
                     'unit': 'TFLOP/s', 'frac': EVAL_FLOPS_PER_POINT * Q * T / (ev_kernel_nohull * 1e-3) / 1e12 / FP64_PEAK_TF,
                     'flops_model': '~3.0 kflop/point at the default order (SURVEY 8d E1)'},
            'points_per_sec_hull_on': Q * T / (ev * 1e-3), 'points_per_sec_hull_off': Q * T / (ev_kernel_nohull * 1e-3),
            'hull_note': 'hull on: the whole vi_eval_f64 call (matrix-core mask pass k_hull_mask_mx + evaluation kernel, HIP events around '
                         'the call); hull off: the evaluation kernel alone'}
        if many is not None:
            out['eval_many_timesteps'] = many
        if batched is not None:
            out['batched_records'] = batched
        for b_ in dq + [dhull, dC, dout]:
            b_.free()
        return out
    return None


def finish(comm, world):
    """Close the process group; exit status 3 when RCCL was to carry the broadcast and did not come up (the line above
    then says comm.rccl_broadcast = false: the parameters travelled over the control socket instead)."""
    comm.close()
    rc = 0
    if world > 1 and comm.backend == 'rccl' and not comm.rccl_ready:
        sys.stderr.write('bench.py: RCCL did not initialise on %d ranks (%s); the shared parameters went over the control '
                         'socket.  VINTERP_DIST_BACKEND=socket runs without RCCL on purpose.\n' % (world, '; '.join(comm.notes)))
        rc = 3
    if any('did not return' in n for n in comm.notes):
        # a communicator bootstrap that never returned leaves a thread inside RCCL: skip the interpreter's and the
        # runtime's finalisers, which could block on it, now that the result line is out
        sys.stderr.flush()
        os._exit(rc)
    return rc


def run_c3(args, ctx, comm, model, h, At, A, R, hull_eq, hull_tol, rank, world):
    """Workload c3 (BASELINE configs[3]): T timesteps sharded ceil(T / world) per rank, fitted as one batch per rank (chi^2
    search, covariance), and EVERY timestep of the shard evaluated on a 256^3 grid by the matrix-core kernel, TILE timesteps
    per call into one tile buffer - fit and evaluation of the whole shard inside the timed region, nothing scaled."""
    from volumetricinterp_amd import _lib, synth
    from volumetricinterp_amd.fitengine import FitEngine
    from volumetricinterp_amd.parallel import shard_bounds
    N, P = model.nbasis, A.shape[0]
    Ttot = int(args.records)
    lo, hi = shard_bounds(Ttot, rank, world)
    share = hi - lo
    resident = os.environ.get('VINTERP_C3_EVAL', 'resident') != 'fused'
    # timesteps per evaluation call: 512 (K2r 64.0 TF against 61.6 at 256 and 57.7 at 128; the tile of densities is 69 GB of the 288)
    TILE = min(512 if resident else 64, max(1, share))
    value, error = synth.synth_records(A, share, seed0=1000 + lo) if share else (np.zeros((0, P)), np.ones((0, P)))
    eng = FitEngine(ctx, At, P, N, {'curvature': R}, ['curvature'])
    eng.upload_records(error**-2., value)
    n = int(args.c3_grid)
    g = synth.query_grid(n)
    Q = g[0].size
    dq = [ctx.to_device(a.ravel()) for a in g]
    dhull = ctx.to_device(hull_eq)
    F = hull_eq.shape[0]
    dC = ctx.empty((max(1, share), N))
    dout = ctx.empty((TILE, Q))                            # one tile of densities, overwritten tile after tile (the consumer's buffer)
    dY = ctx.empty((N, Q)) if resident else None           # the basis matrix of the grid: 19 GB of the GPU's 288
    # the fit's results (coefficients, covariances: 8 N^2 bytes per record, 1.66 GB per 10 000) land in the same page-locked arrays
    # step after step - a consumer that writes a shard out before it fits the next one does the same
    res_bufs = eng.result_buffers(calccov=True) if share else None
    fit_s, eval_ms, basis_ms, step_s_list, outcomes = [], [], [], [], None

    def step(record=False):
        nonlocal outcomes
        t0 = time.perf_counter()
        res = eng.fit_resident([P] * share, calccov=True, out=res_bufs) if share else None
        t1 = time.perf_counter()
        ems = bms = 0.
        if share:
            dC.upload(np.nan_to_num(res['Coeffs']))
            if resident:
                # once per pass: the basis of the grid (K1) with the hull mask folded in as NaN rows ...
                ctx.timer_start()
                _lib.check(_lib.lib.vi_eval_basis_f64(h, Q, dq[0].ptr, dq[1].ptr, dq[2].ptr, dhull.ptr, F, hull_tol, dY.ptr),
                           'vi_eval_basis_f64')
                bms = ctx.timer_stop_ms()
            # ... then EVERY timestep of the shard on the grid, TILE timesteps per call (K2r: one matrix product per call)
            ctx.timer_start()
            for s0 in range(0, share, TILE):
                tl = min(TILE, share - s0)
                if resident:
                    _lib.check(_lib.lib.vi_eval_resident_f64(h, Q, tl, dY.ptr, dC.offset_ptr(s0 * N), dout.ptr),
                               'vi_eval_resident_f64')
                else:
                    _lib.check(_lib.lib.vi_eval_f64(h, Q, dq[0].ptr, dq[1].ptr, dq[2].ptr, tl, dC.offset_ptr(s0 * N), dhull.ptr, F,
                                                    hull_tol, dout.ptr), 'vi_eval_f64')
            ems = ctx.timer_stop_ms()
            outcomes = res['search']['curvature']['outcomes']
        ctx.sync()
        t2 = time.perf_counter()
        if record:
            fit_s.append(t1 - t0)
            eval_ms.append(ems)
            basis_ms.append(bms)
            step_s_list.append(t2 - t0)

    for _ in range(args.warmup):
        step()
    eng.solve_timing(1)             # one HIP event pair around every eigen-solve launch of the timed steps, on its own stream
    ctx.sync()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    ctx.sync()
    comm.barrier()
    wall = comm.max_over_ranks(time.perf_counter() - t0)
    st = eng.solve_timing(0)
    pipelines = eng.stats.get('pipelines', 1)
    # per rank: the measured time of a step - fit of the shard + basis of the grid + evaluation of every timestep of the shard
    mine = float(np.mean(step_s_list)) if share else 0.
    per_rank = [float(np.frombuffer(b_, dtype=np.float64)[0]) for b_ in comm.allgather_bytes(np.array([mine]).tobytes())]
    fits = [float(np.frombuffer(b_, dtype=np.float64)[0])
            for b_ in comm.allgather_bytes(np.array([float(np.mean(fit_s)) if fit_s else 0.]).tobytes())]
    step_s = wall / args.steps            # barrier to barrier, max over ranks: nothing scaled
    # free what the secondary legs of an N = 1 run do not need (the basis matrix of the grid alone is 19 GB)
    for b_ in dq + [dhull, dC, dout] + ([dY] if dY is not None else []):
        b_.free()
    eng.close()
    if rank != 0:
        return None
    # ---- roofline of the dominant part of the step, the FIT (rank 0's shard): the in-LDS eigen-solves K3 - every
    #      k_jacobi_solve launch and the Jacobi rounds inside the k_brent_warm launches (one per pipeline and step: Brent's
    #      iteration of its records, whose whole duration is counted, chi^2 and re-basing included).  Neither HBM nor MFMA
    #      bound them: a system lives in one CU's LDS and every round moves it once through the registers.  Algorithmic
    #      bytes: 2 (read + write) x 8 B x N(N+1)/2 per round and system, rounds counted by the kernels
    #      themselves; algorithmic flops: 12 per stored element and round (two row and two column rotations of 3 flops).
    fit_roof = None
    if st['timed'] and fit_s:
        fit_wall = float(np.mean(fit_s))
        rounds_step = st['rounds'] / args.steps
        launches_step = st['launches'] / args.steps
        avg_ms = st['total_ms'] / st['timed']
        k3_ms_step = avg_ms * launches_step                       # summed over the pipelines' streams (their launches overlap)
        busy_s = min(k3_ms_step * 1e-3, fit_wall)
        lds_bytes_step = 2. * 8. * (N * (N + 1) // 2) * rounds_step
        flops_step = 12. * (N * (N + 1) // 2) * rounds_step
        ach = lds_bytes_step / busy_s / 1e9
        fit_roof = {
            'kernel': 'K3: k_jacobi_solve launches + the Jacobi rounds inside k_brent_warm (in-LDS eigen-solves of the fit)',
            'bound': 'lds', 'achieved': ach, 'peak': LDS_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / LDS_PEAK_GBS,
            'traffic': 8.22e11 * (share / 10000.) if N == 144 else None,
            'traffic_note': 'HBM bytes of the K3 kernels (k_jacobi_solve_v2 + k_brent_warm) per step of 10 000 records from the '
                            'committed PMC passes (profiles/r4_k3_hbm_pmc.txt: FETCH_SIZE as reported + WRITE_SIZE; 1.26e12 with '
                            'every read doubled), scaled by the records of this rank: 0.4-0.6 TB/s, 4-6 % of the bytes through '
                            'LDS - HBM is not the resource',
            'definition': 'algorithmic LDS bytes of all K3 rounds of a step / the time K3 kernels were running in that step = '
                          'min(sum of the launch durations over the %d concurrent pipelines, wall time of the fit)' % pipelines,
            'lds_bytes_per_step': lds_bytes_step, 'rounds_per_step': rounds_step, 'systems_per_step': st['systems'] / args.steps,
            'launches_per_step': launches_step, 'avg_launch_ms': avg_ms, 'max_launch_ms': st['max_ms'],
            'launch_ms_summed_per_step': k3_ms_step, 'fit_wall_ms_per_step': fit_wall * 1e3,
            'share_of_step': fit_wall / step_s,
            'per_launch': {'achieved': lds_bytes_step / max(1e-12, k3_ms_step * 1e-3) / 1e9, 'unit': 'GB/s',
                           'note': 'algorithmic bytes per launch / average launch duration (HIP events on the launching '
                                   'stream) - the figure a rocprofv3 kernel trace reproduces; concurrent launches share the '
                                   'chip, so it understates the chip-wide rate by up to the number of pipelines'},
            'peak_note': 'chip-wide LDS rate for equal read and write volumes: %.0f B/clk/CU x %.1f GHz x %d CUs; one system '
                         'occupies one CU' % (LDS_RW_BYTES_PER_CLK_CU, CLOCK_GHZ, N_CU),
            'fp64': {'achieved_tflops': flops_step / busy_s / 1e12, 'peak_tflops': FP64_PEAK_TF,
                     'frac': flops_step / busy_s / 1e12 / FP64_PEAK_TF,
                     'flops_model': '12 flop per stored element and round, counted rounds (a converged cold solve of 10 sweeps is '
                                    '~30 N^3; SURVEY 8d F2 quotes ~10 N^3 per solve)'},
            'records_per_sec_fit': share / fit_wall}
    k2r_traffic = ({256: K2R_PMC_BYTES_PER_256_TIMESTEPS, 512: K2R_PMC_BYTES_PER_512_TIMESTEPS}.get(TILE)
                   if (resident and Q == 256**3) else None)
    return {
        'metric': 'fit+eval timesteps/sec', 'value': Ttot / step_s, 'unit': 'timesteps/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': step_s * 1e3, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'configs[3]: %d timesteps (26-beam x 100-range records, N=144, curvature, chi2 search, '
                               'covariance) sharded ceil(T/N) per rank, no data-path collective; each rank fits its shard as '
                               'one batch and evaluates it on a 256^3 geodetic grid with the hull mask (%s), fp64'
                               % (Ttot, 'basis matrix of the grid resident in HBM, matrix-core product K2r over up to 512 timesteps per call'
                                  if resident else 'fused matrix-core kernel, basis recomputed per 32 timesteps'),
                   'timesteps': Ttot, 'timesteps_per_rank': -(-Ttot // world), 'grid_points': Q,
                   'evaluation': 'every timestep of the shard is evaluated on the grid inside the timed region, %d timesteps per '
                                 'call into one tile buffer of densities (nothing scaled or skipped); value = timesteps / '
                                 '(barrier-to-barrier wall time per step, max over ranks)' % TILE},
        'points_per_sec': Ttot * Q / step_s,
        'per_rank_step_s': per_rank, 'per_rank_fit_s': fits,
        'rank0': {'fit_s': float(np.mean(fit_s)) if fit_s else None, 'eval_ms': float(np.mean(eval_ms)) if eval_ms else None,
                  'eval_timesteps_per_call': TILE, 'eval_mode': 'resident basis + K2r' if resident else 'fused kernel',
                  'eval_basis_ms_once': float(np.mean(basis_ms)) if basis_ms else None,
                  'eval_basis_bytes': int(N) * int(Q) * 8 if resident else 0,
                  'eval_point_timesteps_per_sec': (share * Q / (float(np.mean(eval_ms)) * 1e-3)) if eval_ms and share else None,
                  'records_per_sec_fit': share / float(np.mean(fit_s)) if fit_s else None,
                  'outcomes': {o_: outcomes.count(o_) for o_ in set(outcomes)} if outcomes else None,
                  'pipelines': pipelines},
        'roofline': fit_roof,
        'roofline_eval': {
            'kernel': ('k_eval_resident (K2r: v_mfma_f64_16x16x4 on the basis matrix of the grid resident in HBM)' if resident
                       else 'k_eval_sph_mfma'),
            'bound': 'mfma', 'achieved': (2. * N * share * Q / (float(np.mean(eval_ms)) * 1e-3) / 1e12) if eval_ms and share else None,
            'peak': FP64_PEAK_TF, 'unit': 'TFLOP/s',
            'frac': (2. * N * share * Q / (float(np.mean(eval_ms)) * 1e-3) / 1e12 / FP64_PEAK_TF) if eval_ms and share else None,
            'ms_per_step': float(np.mean(eval_ms)) if eval_ms else None,
            'share_of_step': (float(np.mean(eval_ms)) * 1e-3 / step_s) if eval_ms else None,
            'flops_model': '2 N flop per point-timestep (SURVEY 8d E2)', 'traffic': k2r_traffic,
            'traffic_note': 'HBM bytes per call of %d timesteps on 256^3 points from the committed PMC passes '
                            '(profiles/r3_k2r_pmc.txt: read + written), algorithmic %.1f GB'
                            % (TILE, (N * Q * 8 + TILE * Q * 8) / 1e9)},
        'comm': {'backend': comm.backend, 'rccl_broadcast': bool(comm.rccl_ready), 'notes': comm.notes},
    }


if __name__ == '__main__':
    main()
