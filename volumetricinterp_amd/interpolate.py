"""Fit the 3-D analytic model to every record of an AMISR file, on the MI355X.

Drop-in mirror of the reference class ``volumetricinterp/interpolate.py:16-708``:
same constructor (``Interpolate(config_file)``), same public methods
(``calc_coeffs, eval_C, find_reg_param, chi2, chi2objfunct, compute_hull,
read_datafile, saveh5``), same result attributes (``time, Coeffs, Covariance,
chi_sq, hull_vert``) and the same failure conventions (NaN rows for records whose
regularisation parameter cannot be found, interpolate.py:142-145, :558-563).

The arithmetic runs in libvinterp.so through ``fitengine.FitEngine``: the record
loop of interpolate.py:511 becomes one batch, see fitengine.py.
"""
import configparser
import datetime as dt
import importlib
import os

import numpy as np
from scipy.spatial import ConvexHull

from . import _lib
from .fitengine import FitEngine
from .geodesy import geodetic2ecef


class Interpolate(object):
    def __init__(self, config_file, ctx=None):
        self.configfile = config_file
        self.read_config(self.configfile)
        self._ctx = ctx
        m = importlib.import_module('.models.' + self.model_name, package='volumetricinterp_amd')
        with open(self.configfile) as f:
            self.model = m.Model(f, ctx=ctx)

    # interpolate.py:64-88
    def read_config(self, config_file):
        config = configparser.ConfigParser()
        with open(config_file) as f:
            config.read_file(f)
        self.regularization_list = list(filter(None, config.get('DEFAULT', 'REGULARIZATION_LIST').split(',')))
        self.reg_method = config.get('DEFAULT', 'REGULARIZATION_METHOD')
        self.filename = config.get('DEFAULT', 'FILENAME')
        self.outputfilename = config.get('DEFAULT', 'OUTPUTFILENAME')
        self.param = config.get('DEFAULT', 'PARAM')
        self.errlim = [float(i) for i in config.get('DEFAULT', 'ERRLIM').split(',')]
        self.chi2lim = [float(i) for i in config.get('DEFAULT', 'CHI2LIM').split(',')]
        self.goodfitcode = [int(i) for i in config.get('DEFAULT', 'GOODFITCODE').split(',')]
        self.model_name = config.get('MODEL', 'NAME')

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.get_context()
        return self._ctx

    # ------------------------------------------------------------------------------------------
    def _check_method(self, method):
        if method is None:
            method = 'chi2'                    # interpolate.py:135-136
        if method in ('chi2', 'gcv'):
            return method
        if method in ('manual', 'prompt'):
            # both raise TypeError in the reference itself (8-argument signatures called with 5)
            raise NotImplementedError("regularisation method %r is broken in the reference and not provided by "
                                      "volumetricinterp_amd ('chi2' and 'gcv' are)" % method)
        raise KeyError(method)

    def _engine_for(self, A, reg_matrices):
        return FitEngine.from_host_basis(self.ctx, A, reg_matrices, self.regularization_list)

    @staticmethod
    def _finite_or_raise(*arrays):
        for a in arrays:
            if not np.all(np.isfinite(a)):
                raise ValueError('array must not contain infs or NaNs')      # scipy.linalg.lstsq check_finite

    # interpolate.py:97-147
    def find_reg_param(self, A, b, W, reg_matrices, method=None):
        method = self._check_method(method)
        b = np.asarray(b, dtype=np.float64).ravel()
        W = np.asarray(W, dtype=np.float64).ravel()
        reg_params = {}
        try:
            self._finite_or_raise(A, b, W)
            eng = self._engine_for(A, reg_matrices)
        except ValueError as err:
            print(err)
            print('Returning NANs for regularization parameters.')
            return {rl: np.nan for rl in self.regularization_list}
        eng.load_records(W[None, :], b[None, :])
        if method == 'gcv':
            params, infos = eng.search_gcv([np.arange(len(b), dtype=np.int32)])
        else:
            params, infos = eng.search([len(b)])
        for rl in self.regularization_list:
            reg_params[rl] = params[0][rl]
            outcome = infos[rl]['outcomes'][0]
            if outcome == 'no_minimum':
                print('Minima of GCV function could not be found')
                print('Returning NANs for regularization parameters.')
            if outcome == 'too_smooth':
                print('Too smooth to find regularization parameter. Returning alpha=0.')
            elif outcome == 'no_root':
                print('Could not find any roots to the objective function chi^2-nu in the range (1e-100,1).')
                print('Returning NANs for regularization parameters.')
        eng.close()
        return reg_params

    # interpolate.py:152-218
    def chi2(self, A, b, W, reg_matrices, reg):
        b = np.asarray(b, dtype=np.float64).ravel()
        W = np.asarray(W, dtype=np.float64).ravel()
        self._finite_or_raise(A, b, W)
        eng = FitEngine.from_host_basis(self.ctx, A, reg_matrices, self.regularization_list)
        eng.load_records(W[None, :], b[None, :])
        from . import alpha_search

        def evaluate(rec, log10a):
            al = {n: (np.power(10., log10a) if n == reg else np.zeros(len(rec))) for n in self.regularization_list}
            return eng.chi2_batch(rec, al)
        alphas, outcomes, _, _ = alpha_search.run_batched([len(b)], evaluate)
        eng.close()
        if outcomes[0] == 'no_root':
            raise ValueError(alpha_search.NO_ROOT_MSG)
        if outcomes[0] == 'too_smooth':
            print('Too smooth to find regularization parameter. Returning alpha=0.')
        return alphas[0]

    # interpolate.py:220-261
    def chi2objfunct(self, alpha, A, b, W, reg_matrices, nu, reg):
        b = np.asarray(b, dtype=np.float64).ravel()
        W = np.asarray(W, dtype=np.float64).ravel()
        self._finite_or_raise(A, b, W)
        eng = FitEngine.from_host_basis(self.ctx, A, reg_matrices, self.regularization_list)
        eng.load_records(W[None, :], b[None, :])
        al = {n: np.array([np.power(10., alpha) if n == reg else 0.]) for n in self.regularization_list}
        c2 = eng.chi2_batch(np.zeros(1, dtype=np.int32), al)[0]
        eng.close()
        return c2 - nu

    # interpolate.py:432-469
    def eval_C(self, A, b, W, reg_matrices, reg_params, calccov=False):
        b = np.asarray(b, dtype=np.float64).ravel()
        W = np.asarray(W, dtype=np.float64).ravel()
        self._finite_or_raise(A, b, W, [reg_params[r] for r in self.regularization_list])
        eng = self._engine_for(A, reg_matrices)
        eng.load_records(W[None, :], b[None, :])
        C, dC, _, _ = eng.finalize([{r: float(reg_params[r]) for r in self.regularization_list}], calccov=calccov)
        eng.close()
        if calccov:
            return C[0], dC[0]
        return C[0]

    # interpolate.py:409-426
    def compute_hull(self, lat, lon, alt):
        x, y, z = geodetic2ecef(lat, lon, alt)
        R_cart = np.array([x, y, z]).T
        chull = ConvexHull(R_cart)
        self.hull_vert = R_cart[chull.vertices]

    # interpolate.py:472-579
    def calc_coeffs(self, starttime=None, endtime=None, comm=None):
        """interpolate.py:472-579.  With a multi-rank `comm` (parallel.Comm; one process per GPU) the records are
        sharded in contiguous blocks: rank 0 evaluates the regularisation matrices and broadcasts them, every rank
        fits its block, and the rows are gathered so that every rank (rank 0 writes the file) holds the full result.
        The record loop of the reference carries no state between records (interpolate.py:511), so the sharded
        result equals the single-process result row for row."""
        multi = comm is not None and comm.world > 1
        reg_matricies = {}
        if not multi or comm.rank == 0:
            print('Evaluating Regularization matricies.  This may take a few minutes.')
            for reg in self.regularization_list:
                try:
                    reg_matricies[reg] = self.model.eval_reg_matricies[reg]()
                except KeyError as e:
                    print('WARNING: The model {} does not support {} regularization!'.format(self.model_name, reg))
                    if multi:
                        comm.broadcast_arrays({'__unsupported__': np.zeros(1)})
                    raise e
        if multi:
            reg_matricies = comm.broadcast_arrays(reg_matricies)
            if '__unsupported__' in reg_matricies:
                raise KeyError('regularization not supported by model {}'.format(self.model_name))

        utime, lat, lon, alt, value, error = self.read_datafile(self.filename)
        self.compute_hull(lat, lon, alt)

        if starttime and endtime:
            t0 = (starttime - dt.datetime(1970, 1, 1)).total_seconds()
            t1 = (endtime - dt.datetime(1970, 1, 1)).total_seconds()
            idx = np.argwhere((utime[:, 0] >= t0) & (utime[:, 1] <= t1)).flatten()
            utime = utime[idx, :]
            value = value[idx]
            error = error[idx]

        if not multi:
            res = self.fit_records(lat, lon, alt, value, error, reg_matricies)
        else:
            from .parallel import shard_bounds
            T, N = value.shape[0], self.model.nbasis
            lo, hi = shard_bounds(T, comm.rank, comm.world)
            names = list(self.regularization_list)
            if hi > lo:
                loc = self.fit_records(lat, lon, alt, value[lo:hi], error[lo:hi], reg_matricies)
                par = np.array([[p[n] for n in names] for p in loc['reg_params']], dtype=np.float64).reshape(hi - lo, len(names))
            else:
                loc = dict(Coeffs=np.zeros((0, N)), Covariance=np.zeros((0, N, N)), chi_sq=np.zeros(0))
                par = np.zeros((0, len(names)))
            res = dict(Coeffs=comm.gather_rows(loc['Coeffs'], T), Covariance=comm.gather_rows(loc['Covariance'], T),
                       chi_sq=comm.gather_rows(np.asarray(loc['chi_sq'], dtype=np.float64).reshape(-1, 1), T)[:, 0])
            if names:
                par = comm.gather_rows(par, T)
                res['reg_params'] = [dict(zip(names, row.tolist())) for row in par]
            else:
                res['reg_params'] = [dict() for _ in range(T)]
        self.time = utime
        self.Coeffs = res['Coeffs']
        self.Covariance = res['Covariance']
        self.chi_sq = res['chi_sq']
        self.reg_params = res['reg_params']

    # interpolate.py:263-297 / :299-351
    def gcv(self, A, b, W, reg_matrices, reg):
        b = np.asarray(b, dtype=np.float64).ravel()
        W = np.asarray(W, dtype=np.float64).ravel()
        self._finite_or_raise(A, b, W)
        eng = FitEngine.from_host_basis(self.ctx, A, reg_matrices, self.regularization_list)
        eng.load_records(W[None, :], b[None, :])
        saved = eng.regularization_list
        eng.regularization_list = [reg]
        params, infos = eng.search_gcv([np.arange(len(b), dtype=np.int32)])
        eng.regularization_list = saved
        eng.close()
        if infos[reg]['outcomes'][0] != 'minimum':
            raise ValueError('Minima of GCV function could not be found')
        return params[0][reg]

    def gcvobjfunct(self, alpha, A0, b0, W0, reg_matrices, reg):
        b0 = np.asarray(b0, dtype=np.float64).ravel()
        W0 = np.asarray(W0, dtype=np.float64).ravel()
        self._finite_or_raise(A0, b0, W0)
        eng = FitEngine.from_host_basis(self.ctx, A0, reg_matrices, self.regularization_list)
        eng.load_records(W0[None, :], b0[None, :])
        v = eng.gcv_objective(0, alpha, reg, np.arange(len(b0), dtype=np.int32))
        eng.close()
        return v

    def fit_records(self, lat, lon, alt, value, error, reg_matricies, calccov=True, record_slice=None):
        """The record loop of interpolate.py:511-579 as one device batch.

        value, error: (T, P) arrays sharing the coordinates lat/lon/alt (P,)."""
        lat = np.asarray(lat, dtype=np.float64).ravel()
        lon = np.asarray(lon, dtype=np.float64).ravel()
        alt = np.asarray(alt, dtype=np.float64).ravel()
        value = np.atleast_2d(np.asarray(value, dtype=np.float64))
        error = np.atleast_2d(np.asarray(error, dtype=np.float64))
        T, P = value.shape
        N = self.model.nbasis
        ctx = self.ctx
        self.model.handle(ctx)
        dlat, dlon, dalt = ctx.to_device(lat), ctx.to_device(lon), ctx.to_device(alt)
        At = self.model.basis_device(dlat, dlon, dalt, P, transposed=True)
        basis_ok = bool(np.all(np.isfinite(At.download())))

        # points with non-finite value are dropped (mask derives from the value only, interpolate.py:516-520)
        fin = np.isfinite(value)
        with np.errstate(all='ignore'):
            W = np.where(fin, error**(-2), 0.)
        b = np.where(fin, value, 0.)
        rec_ok = np.all(np.isfinite(W), axis=1) & basis_ok
        if not np.all(rec_ok):
            if not self.regularization_list:
                # eval_C is reached directly and scipy.linalg.lstsq raises (uncaught in the reference)
                raise ValueError('array must not contain infs or NaNs')
            print('array must not contain infs or NaNs')
            print('Returning NANs for regularization parameters.')
            W[~rec_ok] = 0.
        npts = [int(n) if ok else None for n, ok in zip(fin.sum(axis=1), rec_ok)]

        method = self._check_method(self.reg_method)
        point_lists = [np.nonzero(fin[t])[0].astype(np.int32) if npts[t] is not None else None for t in range(T)]
        eng = FitEngine(ctx, At, P, N, reg_matricies, self.regularization_list)
        try:
            res = eng.fit(W, b, npts, calccov=calccov, method=method, point_lists=point_lists)
        finally:
            self.fit_stats = dict(eng.stats)
            eng.close()
        return res

    # interpolate.py:582-667
    def read_datafile(self, filename):
        from .h5io import read_amisr_file
        return read_amisr_file(filename, self.param, self.errlim, self.chi2lim, self.goodfitcode)

    # interpolate.py:671-708
    def saveh5(self):
        from .h5io import write_coeff_file
        Path = os.path.dirname(os.path.abspath(self.configfile))
        Name = os.path.basename(self.configfile)
        with open(self.configfile, 'r') as f:
            Contents = ''.join(f.readlines())
        write_coeff_file(self.outputfilename, time=self.time, Coeffs=self.Coeffs, Covariance=self.Covariance,
                         reglist=self.regularization_list, regmethod=self.reg_method, chi2=self.chi_sq,
                         hull_vert=self.hull_vert, rawfilename=self.filename, config_name=Name, config_path=Path,
                         config_contents=Contents)
