"""Evaluate a fitted model at arbitrary geodetic points on the MI355X.

Drop-in mirror of the reference class ``volumetricinterp/estimate.py:13-221``:
same constructor, same ``__call__`` signature and error behaviour, same
``get_C`` time selection.  ``__call__`` runs the fused HIP kernel
(``vi_eval_f64``: coordinates -> basis -> contraction with C -> hull mask)
instead of materialising ``A`` (1 152 B per point at N = 144) and calling
``einsum`` (estimate.py:113-115); the per-point Qhull of ``check_hull``
(estimate.py:153-178) becomes one host Qhull plus a half-space test fused in
the kernel.
"""
import configparser
import datetime as dt
import importlib
import io

import numpy as np
from scipy.spatial import ConvexHull

from . import _lib


def hull_equations(hull_vert):
    """Facet half-spaces (F, 4) of the data hull: inside <=> eq[:, :3] @ x + eq[:, 3] <= tol."""
    hull_vert = np.ascontiguousarray(hull_vert, dtype=np.float64)
    eq = np.ascontiguousarray(ConvexHull(hull_vert).equations, dtype=np.float64)
    # Qhull treats a point within its distance round-off of a facet as coplanar (not a new vertex),
    # which the reference's vertex-list comparison (estimate.py:174-176) then reports as inside.
    tol = 4. * np.finfo(np.float64).eps * 3. * float(np.max(np.abs(hull_vert)))
    return order_facets(eq, hull_vert), tol


def order_facets(eq, hull_vert, nsample=4096, nlead=32):
    """The same half-spaces, the most telling ones first.  The mask pass (k_hull_mask) takes the maximum over all facets - the
    order cannot change a result - but leaves as soon as every point of a wave has been found outside by SOME facet: with
    Qhull's order an outside point met its first violated facet late.  Greedy cover on a sample of the vertices' bounding
    box: the facet that rejects most sample points first, then the one that rejects most of the rest, ... (nlead of them);
    the others keep their order."""
    if len(eq) <= nlead:
        return eq
    lo, hi = hull_vert.min(axis=0), hull_vert.max(axis=0)
    pts = lo + (hi - lo) * np.random.default_rng(0).random((nsample, 3))
    out = (pts @ eq[:, :3].T + eq[:, 3]) > 0.                     # (sample, facet): the facet puts the point outside
    left = out.any(axis=1)
    lead = []
    for _ in range(nlead):
        if not left.any():
            break
        gain = out[left].sum(axis=0)
        gain[lead] = -1
        f = int(np.argmax(gain))
        if gain[f] <= 0:
            break
        lead.append(f)
        left &= ~out[:, f]
    rest = np.setdiff1d(np.arange(len(eq)), lead)
    return np.ascontiguousarray(eq[np.concatenate([np.array(lead, dtype=np.int64), rest])])


class Estimate(object):
    def __init__(self, coeff_filename, timetol=60., timeinterp=False, ctx=None):
        self.timetol = timetol
        self.timeinterp = timeinterp
        self._ctx = ctx
        self.loadh5(filename=coeff_filename)
        self._init_model()

    @classmethod
    def from_arrays(cls, Coeffs, Covariance, time, hull_vert, config_text, timetol=60., timeinterp=False, ctx=None):
        """Build an Estimate from in-memory fit results (what loadh5 would have read)."""
        self = cls.__new__(cls)
        self.timetol, self.timeinterp, self._ctx = timetol, timeinterp, ctx
        self.Coeffs = np.asarray(Coeffs, dtype=np.float64)
        self.Covariance = None if Covariance is None else np.asarray(Covariance, dtype=np.float64)
        self.time = np.asarray(time)
        self.hull_vert = np.asarray(hull_vert, dtype=np.float64)
        self.config_file_text = config_text.encode('utf-8') if isinstance(config_text, str) else config_text
        self._init_model()
        return self

    def _init_model(self):
        # estimate.py:41-50: the model is rebuilt from the config text embedded in the file
        config_file = io.StringIO(self.config_file_text.decode('utf-8'))
        config = configparser.ConfigParser()
        config.read_file(config_file)
        self.model_name = config.get('MODEL', 'NAME')
        config_file.seek(0)
        m = importlib.import_module('.models.' + self.model_name, package='volumetricinterp_amd')
        self.model = m.Model(config_file, ctx=self._ctx)
        self._hull_eq = None

    # estimate.py:53-70
    def loadh5(self, filename=None):
        from .h5io import read_coeff_file
        d = read_coeff_file(filename)
        self.Coeffs = d['Coeffs']
        self.Covariance = d['Covariance']
        self.time = d['time']
        self.hull_vert = d['hull_vert']
        self.config_file_text = d['config_file_text']

    def _hull(self):
        if self._hull_eq is None:
            self._hull_eq = hull_equations(self.hull_vert)
        return self._hull_eq

    # estimate.py:75-123
    def __call__(self, time, gdlat, gdlon, gdalt, calcgrad=False, calcerr=False, check_hull=True):
        # calcgrad / calcerr are accepted and ignored, exactly as in the reference (code after the
        # `return` at estimate.py:123 is dead)
        C, dC = self.get_C(time)
        gdlat = np.asarray(gdlat, dtype=np.float64)
        out = self.evaluate_coeffs(np.asarray(C, dtype=np.float64)[None, :], gdlat, gdlon, gdalt, check_hull)
        return out[0].reshape(gdlat.shape)

    def evaluate_coeffs(self, C, gdlat, gdlon, gdalt, check_hull=True, out=None):
        """out[t] = density of coefficient row C[t] at the points; (T, Q).  `out`: optional C-contiguous float64 (T, Q)
        array to write into (e.g. from _lib.pinned_empty, like the coordinate arrays, for full-rate transfers)."""
        C = np.ascontiguousarray(C, dtype=np.float64)
        lat = np.ascontiguousarray(np.asarray(gdlat, dtype=np.float64).ravel())
        lon = np.ascontiguousarray(np.asarray(gdlon, dtype=np.float64).ravel())
        alt = np.ascontiguousarray(np.asarray(gdalt, dtype=np.float64).ravel())
        if not (lat.size == lon.size == alt.size):
            raise ValueError('gdlat, gdlon, gdalt must have the same shape')
        T, Q = C.shape[0], lat.size
        if C.shape[1] != self.model.nbasis:
            raise ValueError('coefficient vector length %d != nbasis %d' % (C.shape[1], self.model.nbasis))
        if out is None:
            out = np.empty((T, Q), dtype=np.float64)
        elif out.shape != (T, Q) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError('out must be a C-contiguous float64 array of shape (%d, %d)' % (T, Q))
        if Q == 0 or T == 0:
            return out
        h = self.model.handle()
        P = _lib.c_double_p
        if check_hull:
            eq, tol = self._hull()
            F, eqp = eq.shape[0], eq.ctypes.data_as(P)
        else:
            tol, F, eqp = 0., 0, None
        _lib.check(_lib.lib.vi_eval_f64_host(h, Q, lat.ctypes.data_as(P), lon.ctypes.data_as(P),
                                             alt.ctypes.data_as(P), T, C.ctypes.data_as(P), eqp, F, tol,
                                             out.ctypes.data_as(P)), 'vi_eval_f64_host')
        return out

    def resident_grid(self, gdlat, gdlon, gdalt, check_hull=True):
        """The points of a grid that MANY timesteps are going to be evaluated on (the reference calls Estimate.__call__ once
        per timestep and rebuilds the basis of the grid every time, estimate.py:110-115): their basis matrix is assembled once
        and stays in device memory (N x Q doubles - 19 GB at the default order on a 256^3 grid), and every batch of timesteps
        is one matrix product (ResidentGrid.evaluate_coeffs / ResidentGrid.__call__).  Same values as __call__ to rounding."""
        return ResidentGrid(self, gdlat, gdlon, gdalt, check_hull)

    def gradient(self, time, gdlat, gdlon, gdalt, check_hull=True):
        """Gradient of the fitted parameter at the points: array of shape gdlat.shape + (3,), components along the
        model coordinates z, theta, phi exactly as ``Model.grad_basis`` defines them (sphharmlag.py:148-184), NaN outside
        the hull.  This is the output the reference's ``__call__`` advertises as ``calcgrad`` but never computes (the
        code after its ``return`` is dead, estimate.py:125-147); ``__call__`` itself keeps ignoring the flag,
        as the reference does."""
        C, dC = self.get_C(time)
        gdlat = np.asarray(gdlat, dtype=np.float64)
        lat = np.ascontiguousarray(gdlat.ravel())
        lon = np.ascontiguousarray(np.asarray(gdlon, dtype=np.float64).ravel())
        alt = np.ascontiguousarray(np.asarray(gdalt, dtype=np.float64).ravel())
        if not (lat.size == lon.size == alt.size):
            raise ValueError('gdlat, gdlon, gdalt must have the same shape')
        Q = lat.size
        out = np.empty((Q, 3))
        if Q:
            h = self.model.handle()
            ctx = self.model._ctx
            bufs = []                               # freed whatever happens (a failed call must not keep device memory)
            try:
                for a in (lat, lon, alt):
                    bufs.append(ctx.to_device(a))
                bufs.append(ctx.to_device(np.ascontiguousarray(C, dtype=np.float64)))
                bufs.append(ctx.empty((Q, 3)))
                _lib.check(_lib.lib.vi_eval_grad_f64(h, Q, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr),
                           'vi_eval_grad_f64')
                out = bufs[4].download()
            finally:
                for a in bufs:
                    a.free()
            if check_hull:
                out[~self.check_hull(lat, lon, alt)] = np.nan
        return out.reshape(gdlat.shape + (3,))

    def error(self, time, gdlat, gdlon, gdalt, check_hull=True):
        """Standard error of the fitted parameter at the points, sqrt(a^T dC a) with a the basis row of the point and dC
        the coefficient covariance of the record (first-order error propagation): the ``calcerr`` output the reference
        advertises but never computes (estimate.py:139-145).  Same shape as gdlat, NaN outside the hull."""
        C, dC = self.get_C(time)
        gdlat = np.asarray(gdlat, dtype=np.float64)
        lat = np.ascontiguousarray(gdlat.ravel())
        lon = np.ascontiguousarray(np.asarray(gdlon, dtype=np.float64).ravel())
        alt = np.ascontiguousarray(np.asarray(gdalt, dtype=np.float64).ravel())
        if not (lat.size == lon.size == alt.size):
            raise ValueError('gdlat, gdlon, gdalt must have the same shape')
        Q = lat.size
        out = np.empty(Q)
        if Q:
            h = self.model.handle()
            ctx = self.model._ctx
            bufs = []
            try:
                for a in (lat, lon, alt):
                    bufs.append(ctx.to_device(a))
                bufs.append(ctx.to_device(np.ascontiguousarray(dC, dtype=np.float64)))
                bufs.append(ctx.empty((Q,)))
                _lib.check(_lib.lib.vi_eval_err_f64(h, Q, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr),
                           'vi_eval_err_f64')
                out = bufs[4].download()
            finally:
                for a in bufs:
                    a.free()
            if check_hull:
                out[~self.check_hull(lat, lon, alt)] = np.nan
        return out.reshape(gdlat.shape)

    # estimate.py:153-178 (boolean mask, same shape as the inputs)
    def check_hull(self, lat0, lon0, alt0):
        alt0 = np.asarray(alt0, dtype=np.float64)
        z = np.zeros((1, self.model.nbasis))
        out = self.evaluate_coeffs(z, lat0, lon0, alt0, check_hull=True)
        return np.isfinite(out[0]).reshape(alt0.shape)

    # estimate.py:180-221
    def get_C(self, t):
        t0 = (t - dt.datetime(1970, 1, 1)).total_seconds()
        mt = np.mean(self.time, axis=1)
        try:
            if self.timeinterp:
                i = np.argwhere((t0 >= mt[:-1]) & (t0 < mt[1:])).flatten()[0]
                T = (t0 - mt[i]) / (mt[i + 1] - mt[i])
                C = (1 - T) * self.Coeffs[i, :] + T * self.Coeffs[i + 1, :]
                dC = (1 - T) * self.Covariance[i, :, :] + T * self.Covariance[i + 1, :, :]
            else:
                i = np.argmin(np.abs(mt - t0))
                if np.abs(mt[i] - t0) > self.timetol:
                    raise IndexError
                C = self.Coeffs[i]
                dC = self.Covariance[i]
        except IndexError:
            raise ValueError('Requested time out of range of data file.')
        return C, dC


class ResidentGrid(object):
    """Basis matrix of a fixed set of points, resident on the device (Estimate.resident_grid)."""

    def __init__(self, est, gdlat, gdlon, gdalt, check_hull=True):
        self.est = est
        self.shape = np.asarray(gdlat).shape
        lat = np.ascontiguousarray(np.asarray(gdlat, dtype=np.float64).ravel())
        lon = np.ascontiguousarray(np.asarray(gdlon, dtype=np.float64).ravel())
        alt = np.ascontiguousarray(np.asarray(gdalt, dtype=np.float64).ravel())
        if not (lat.size == lon.size == alt.size):
            raise ValueError('gdlat, gdlon, gdalt must have the same shape')
        self.Q = lat.size
        ctx = est.model.ctx
        N = est.model.nbasis
        free, _ = ctx.mem_info()
        if self.Q * N * 8 > 0.9 * free:
            raise MemoryError('basis matrix of %d points x %d functions (%.1f GB) does not fit the device (%.1f GB free)'
                              % (self.Q, N, self.Q * N * 8 / 1e9, free / 1e9))
        self.dY = None
        tmp = []                        # device temporaries of the set-up: freed whatever happens below
        try:
            self.dY = ctx.empty((N, self.Q))
            if self.Q == 0:
                return
            for a in (lat, lon, alt):
                tmp.append(ctx.to_device(a))
            if check_hull:
                eq, tol = est._hull()
                dh, F = ctx.to_device(eq), eq.shape[0]
                tmp.append(dh)
            else:
                dh, F, tol = None, 0, 0.
            _lib.check(_lib.lib.vi_eval_basis_f64(est.model.handle(), self.Q, tmp[0].ptr, tmp[1].ptr, tmp[2].ptr,
                                                  dh.ptr if dh is not None else None, F, tol, self.dY.ptr), 'vi_eval_basis_f64')
            ctx.sync()
        except BaseException:
            self.close()                # the basis matrix (19 GB at the default order on 256^3) must not outlive a failed set-up
            raise
        finally:
            for a in tmp:
                a.free()

    def evaluate_coeffs(self, C, out=None):
        """out[t] = density of coefficient row C[t] on the grid; (T, Q) host array."""
        C = np.ascontiguousarray(C, dtype=np.float64)
        N = self.est.model.nbasis
        if C.ndim != 2 or C.shape[1] != N:
            raise ValueError('coefficients must have shape (T, %d)' % N)
        T = C.shape[0]
        if out is None:
            out = np.empty((T, self.Q), dtype=np.float64)
        elif out.shape != (T, self.Q) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError('out must be a C-contiguous float64 array of shape (%d, %d)' % (T, self.Q))
        if T == 0 or self.Q == 0:
            return out
        ctx = self.est.model.ctx
        # timesteps in slabs whose output fits a quarter of the free device memory
        free, _ = ctx.mem_info()
        slab = int(max(1, min(T, (free // 4) // max(1, self.Q * 8))))
        if self.dY is None:
            raise ValueError('this ResidentGrid has been closed')
        dC = dO = None
        try:
            dC = ctx.to_device(C)
            dO = ctx.empty((slab, self.Q))
            for t0 in range(0, T, slab):
                tc = min(slab, T - t0)
                _lib.check(_lib.lib.vi_eval_resident_f64(self.est.model.handle(), self.Q, tc, self.dY.ptr, dC.offset_ptr(t0 * N),
                                                         dO.ptr), 'vi_eval_resident_f64')
                _lib.check(_lib.lib.vi_d2h(ctx.handle, out[t0:t0 + tc].ctypes.data_as(_lib.VOIDP), dO.ptr, tc * self.Q * 8), 'd2h')
        finally:
            for a in (dC, dO):
                if a is not None:
                    a.free()
        return out

    def __call__(self, times):
        """Densities at the grid for a list of datetimes (Estimate.get_C per time): array (len(times),) + grid shape."""
        C = np.array([np.asarray(self.est.get_C(t)[0], dtype=np.float64) for t in times])
        return self.evaluate_coeffs(C).reshape((len(times),) + tuple(self.shape))

    def close(self):
        """Give the basis matrix back to the device (idempotent).  Also runs on `with est.resident_grid(...) as g:` exit
        and when the object is collected."""
        dY, self.dY = getattr(self, 'dY', None), None
        if dY is not None:
            dY.free()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:               # interpreter shutdown: the library may already be gone
            pass
