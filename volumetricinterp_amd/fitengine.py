"""Batched regularised least-squares fit on the MI355X.

Device-side engine behind ``Interpolate.calc_coeffs`` / ``eval_C`` /
``find_reg_param`` (reference: volumetricinterp/interpolate.py:97-261, :432-469,
:511-579).  What changes with respect to the reference's serial record loop:

* the basis matrix is built once per file on the device (N x P layout) and is
  shared by every record; dropped points (non-finite value, interpolate.py:516-520)
  enter with W = 0, b = 0, which removes exactly the same terms from every sum;
* A^T W A and A^T W b are formed once per record (``vi_normal_eq_f64``) instead
  of once per trial alpha (the reference rebuilds them 367-491 times per record);
* all (record, alpha) systems requested in one step of the search are formed,
  eigen-decomposed, truncated and scored in one batch
  (``vi_form_system_f64`` -> ``vi_solve_trunc_f64`` -> ``vi_chi2_f64``);
* the systems of a search resemble each other, and the engine uses it (DESIGN.md
  sections 4-5): Brent's iterates are solved in a rotated system of the record that
  is moved next to the root once they cluster; a batch solves its bracket walk in
  the eigenbases of one reference system per decade (values used for signs only,
  the bracket ends come from cold solves); walk decades whose systems are one and
  the same matrix are solved once; a record fitted alone decomposes the candidate
  bracket bases in the launch of its walk; a big batch runs as concurrent pipelines.
  None of this changes a record's numbers with the batch it is fitted in, bit for bit.
"""
import ctypes as C
import math
import os
import threading

import numpy as np

from . import _lib
from . import alpha_search

EPS = float(np.finfo(np.float64).eps)

_lib._sig('vi_normal_eq_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP,
          _lib.VOIDP, _lib.VOIDP, _lib.VOIDP)
_lib._sig('vi_form_system_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          _lib.VOIDP, _lib.VOIDP)
_lib._sig('vi_solve_trunc_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          C.c_double, _lib.VOIDP, _lib.VOIDP, C.c_double, _lib.VOIDP)
_lib._sig('vi_chi2_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          _lib.VOIDP, _lib.VOIDP, _lib.VOIDP)
_lib._sig('vi_cov_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP)
_lib._sig('vi_warm_prepare_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          _lib.VOIDP, C.c_double, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP)
_lib._sig('vi_warm_solve_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          _lib.VOIDP, _lib.VOIDP, C.c_double, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP)
_lib._sig('vi_basis_solve_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, C.c_double, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP)
_lib._sig('vi_max_sweeps', C.c_int)
_lib._sig('vi_brent_warm_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, C.c_int64, *([_lib.VOIDP] * 18), C.c_double,
          *([_lib.VOIDP] * 5))
_lib._sig('vi_exp10_f64', C.c_int, _lib.VOIDP, _lib.VOIDP, C.c_int64)
_lib._sig('vi_reg_floor_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP)
_lib._sig('vi_rotation_log_bytes', C.c_size_t, C.c_int32)
_lib._sig('vi_decompose_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          _lib.VOIDP, C.c_double, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP)
_lib._sig('vi_warm_finish_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP)
_lib._sig('vi_warm_rebase_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          _lib.VOIDP, _lib.VOIDP, C.c_double, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          _lib.VOIDP)
_lib._sig('vi_warm_chi2_one_f64', C.c_int, _lib.VOIDP, C.c_int32, C.c_int64, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          C.c_int32, C.c_double, C.c_double, _lib.VOIDP, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          C.POINTER(C.c_double))
_lib._sig('vi_gcv_terms_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, C.c_double, _lib.VOIDP, C.c_double, _lib.VOIDP)
_lib._sig('vi_eigvals_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP)
_lib._sig('vi_qr_similarity_f64', C.c_int, _lib.VOIDP, C.c_int64, C.c_int32, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP, _lib.VOIDP,
          _lib.VOIDP)
_lib._sig('vi_brent_host_one_f64', C.c_int, _lib.VOIDP, C.c_int32, C.c_int64, *([_lib.VOIDP] * 11), C.c_int32, C.c_int32,
          C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _lib.VOIDP, _lib.VOIDP)
_lib._sig('vi_brent_warm_supported', C.c_int, C.c_int32, C.c_int64)
_lib.EXPORTS += ['vi_brent_warm_supported', 'vi_brent_host_one_f64', 'vi_brent_warm_f64', 'vi_exp10_f64', 'vi_max_sweeps', 'vi_qr_similarity_f64', 'vi_rotation_log_bytes', 'vi_decompose_f64', 'vi_warm_finish_f64', 'vi_warm_rebase_f64', 'vi_reg_floor_f64', 'vi_basis_solve_f64', 'vi_warm_chi2_one_f64', 'vi_gcv_terms_f64', 'vi_warm_prepare_f64', 'vi_warm_solve_f64', 'vi_eigvals_f64', 'vi_normal_eq_f64', 'vi_form_system_f64', 'vi_solve_trunc_f64', 'vi_chi2_f64', 'vi_cov_f64']

MAX_BATCH = 8192          # systems per solver launch (N=144: 1.3 GB of X)


class _View(object):
    """A window into another engine's device buffer (records lo.. of a batch), for the pipelines of fit_resident."""

    def __init__(self, parent, offset_elems):
        self._parent = parent                   # keeps the allocation alive
        self.dtype = parent.dtype
        self.ptr = parent.offset_ptr(offset_elems)

    def offset_ptr(self, nelem):
        return C.c_void_p(self.ptr.value + int(nelem) * self.dtype.itemsize)


class FitEngine(object):
    # device buffers that hold a record set between calls; every other buffer is scratch of the fit in progress
    RESIDENT_BUFFERS = ('W', 'b', 'Wref', 'bref', 'AWA', 'y')

    def __init__(self, ctx, At_dev, P, N, reg_matrices, regularization_list, scratch_of=None):
        """scratch_of: another engine of the same context whose scratch buffers this one uses (engines that hold
        different record sets of one geometry and are never fitted at the same time - bench.py's sixteen records)."""
        self._scratch_of = scratch_of
        self.ctx = ctx
        self.At = At_dev
        self.P, self.N = int(P), int(N)
        self.regularization_list = list(regularization_list)
        self.R = {}
        self._R_host = {}
        for name in self.regularization_list:
            M = np.ascontiguousarray(reg_matrices[name], dtype=np.float64)
            if M.shape != (self.N, self.N):
                raise ValueError('regularisation matrix %r has shape %r, expected %r' % (name, M.shape, (N, N)))
            if not np.all(np.isfinite(M)):
                raise ValueError('array must not contain infs or NaNs')
            self.R[name] = ctx.to_device(M)
            self._R_host[name] = M
        self._bufs = {}
        self.T = 0
        self._ref_rec = None
        self._subs = None
        self._bounds = [0, 0]
        self._same_below = {}
        self._walk_cache = {}
        self._warm_slot = {}
        self._spec_slot = {}
        self._basis_x = {}
        self.stats = dict(solves=0, launches=0)

    @classmethod
    def from_host_basis(cls, ctx, A, reg_matrices, regularization_list):
        A = np.asarray(A, dtype=np.float64)
        if A.ndim != 2:
            raise ValueError('A must be (npoints, nbasis)')
        if not np.all(np.isfinite(A)):
            raise ValueError('array must not contain infs or NaNs')
        At = ctx.to_device(np.ascontiguousarray(A.T))
        return cls(ctx, At, A.shape[0], A.shape[1], reg_matrices, regularization_list)

    def _buf(self, name, shape, dtype=np.float64):
        if self._scratch_of is not None and name not in self.RESIDENT_BUFFERS:
            return self._scratch_of._buf(name, shape, dtype)
        n = int(np.prod(shape, dtype=np.int64))
        cur = self._bufs.get(name)
        if cur is None or cur.size < n or cur.dtype != np.dtype(dtype):
            if cur is not None:
                cur.free()
            cur = _lib.DeviceArray(self.ctx, (max(n, 1),), dtype)
            self._bufs[name] = cur
        return cur

    # ------------------------------------------------------------------------------------------
    def upload_records(self, W, b):
        """Make weights / data of T records (T, P) resident on the device."""
        W = np.ascontiguousarray(W, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        if W.shape != b.shape or W.ndim != 2 or W.shape[1] != self.P:
            raise ValueError('W, b must both be (T, %d)' % self.P)
        self.T = T = W.shape[0]
        # the pipeline sub-engines (and their contexts: stream, rocBLAS handle, events, workspace) are kept: they adopt
        # the new views in _fit_pipelined; if the number of pipelines changes, _fit_pipelined closes them first
        self.dW = self._buf('W', W.shape).upload(W) if T else None
        self.db = self._buf('b', b.shape).upload(b) if T else None
        K = self.pipelines()
        self._bounds = [(T * k) // K for k in range(K + 1)]
        # the reference system(s) of the shared walk: the mean weights of the batch (of each pipeline's share of it)
        self._ref_host = [(np.mean(W[lo:hi], axis=0, keepdims=True), np.mean(b[lo:hi], axis=0, keepdims=True))
                          for lo, hi in zip(self._bounds[:-1], self._bounds[1:])] if T else []
        self._ref_rec = None
        if T:
            self._set_reference(*(self._ref_host[0] if K == 1 else (np.mean(W, axis=0, keepdims=True),
                                                                    np.mean(b, axis=0, keepdims=True))))

    def _set_reference(self, Wref, bref):
        """One extra 'record' T: mean weights - the reference system whose eigenbases, decade by decade, the bracket walk
        of every record is solved in (chi2_batch_search)."""
        self._ref_rec = None
        if self.T and self.shared_walk_enabled():
            self.dWref = self._buf('Wref', (1, self.P)).upload(Wref)
            self.dbref = self._buf('bref', (1, self.P)).upload(bref)
            self._ref_rec = self.T

    def adopt_records(self, dW, db, T, Wref, bref):
        """Use T records already resident on the device (views into another engine's buffers)."""
        self.T = int(T)
        self._close_subs()
        self.dW, self.db = dW, db
        self._bounds = [0, self.T]
        self._ref_host = [(Wref, bref)]
        self._set_reference(Wref, bref)

    PIPELINE_MIN_RECORDS = 75     # measured: 300 records 376 ms in one pipeline, 327 in four; 100 records 224 / 204; 32: no gain

    def pipelines(self):
        """Number of concurrent fit pipelines a batch is split into (fit_resident)."""
        if getattr(self, '_no_pipeline', False) or not self.warm_enabled():
            return 1
        k = os.environ.get('VINTERP_PIPELINES')
        if k is not None:
            return max(1, min(int(k), max(1, self.T)))
        # four pipelines need more hardware queues than the runtime's default of four (_lib asks for eight at import;
        # with fewer - an explicit GPU_MAX_HW_QUEUES in the environment - two pipelines are what was measured to pay)
        try:
            queues = int(_lib.HW_QUEUES_REQUESTED)
        except ValueError:
            queues = 4
        self.stats['hw_queues_requested'] = queues
        return int(max(1, min(4 if queues >= 8 else 2, self.T // self.PIPELINE_MIN_RECORDS)))

    def form_normal_equations(self):
        """A^T W A (T,N,N) and A^T W b (T,N) of the resident records - once per record, not per alpha."""
        T, N = self.T, self.N
        self.dAWA = self._buf('AWA', (T + 1, N, N))
        self.dy = self._buf('y', (T + 1, N))
        if T:
            _lib.check(_lib.lib.vi_normal_eq_f64(self.ctx.handle, T, self.P, N, self.At.ptr, self.dW.ptr, self.db.ptr,
                                                 self.dAWA.ptr, self.dy.ptr), 'vi_normal_eq_f64')
            if self._ref_rec is not None:
                _lib.check(_lib.lib.vi_normal_eq_f64(self.ctx.handle, 1, self.P, N, self.At.ptr, self.dWref.ptr,
                                                     self.dbref.ptr, self.dAWA.offset_ptr(T * N * N),
                                                     self.dy.offset_ptr(T * N)), 'vi_normal_eq_f64')

    def load_records(self, W, b):
        self.upload_records(W, b)
        self.form_normal_equations()

    def normal_equations(self):
        """Host copies of A^T W A (T,N,N) and A^T W b (T,N) (for stage-wise parity tests)."""
        T, N = self.T, self.N
        AWA = np.empty((T, N, N))
        y = np.empty((T, N))
        _lib.check(_lib.lib.vi_d2h(self.ctx.handle, AWA.ctypes.data_as(_lib.VOIDP), self.dAWA.ptr, AWA.nbytes), 'd2h')
        _lib.check(_lib.lib.vi_d2h(self.ctx.handle, y.ctypes.data_as(_lib.VOIDP), self.dy.ptr, y.nbytes), 'd2h')
        return AWA, y

    def _solve_chunk(self, rec, alphas, want_H, tag):
        """Form and solve B systems; returns device C (B,N) [and H (B,N,N)]."""
        B, N = len(rec), self.N
        drec = self._buf(tag + 'rec', (B,), np.int32).upload(rec)
        dX = self._buf(tag + 'X', (B, N, N))
        first = True
        for name in self.regularization_list:
            a = np.ascontiguousarray(alphas[name], dtype=np.float64)
            dal = self._buf(tag + 'alpha_' + name, (B,)).upload(a)
            _lib.check(_lib.lib.vi_form_system_f64(self.ctx.handle, B, N, self.dAWA.ptr if first else None,
                                                   drec.ptr, dal.ptr, self.R[name].ptr, dX.ptr), 'vi_form_system_f64')
            first = False
        if first:       # no regularisation at all (radbasfun: REGULARIZATION_LIST empty)
            _lib.check(_lib.lib.vi_form_system_f64(self.ctx.handle, B, N, self.dAWA.ptr, drec.ptr, None, None, dX.ptr),
                       'vi_form_system_f64')
        dC = self._buf(tag + 'C', (B, N))
        drank = self._buf(tag + 'rank', (B,), np.int32)
        dH = self._buf(tag + 'H', (B, N, N)) if want_H else None
        # lstsq: rcond = eps (interpolate.py:462); pinv: rtol = max(M,N) * eps (interpolate.py:465)
        _lib.check(_lib.lib.vi_solve_trunc_f64(self.ctx.handle, B, N, dX.ptr, self.dy.ptr, drec.ptr, EPS, dC.ptr,
                                               drank.ptr, N * EPS, dH.ptr if want_H else None), 'vi_solve_trunc_f64')
        self.stats['solves'] += B
        self.stats['launches'] += 1
        return drec, dC, dH, drank

    def _max_batch(self):
        """Systems per solver launch: MAX_BATCH, or fewer where the systems themselves would exceed 4 GiB (N = 1152, the
        doubled order of BASELINE configs[4]: 10.6 MB per system -> 404 per launch; rocSOLVER's batched syevd costs the same per
        system from 64 systems on).  Measured, round 4: ONE launch of 1618 systems at that order (17 GB, 2.147e9 elements - a
        hair under 2^31) ended in a GPU memory access fault inside the library call; launches of up to 392 systems (4.2 GB) are
        what rounds 1-3 ran and tested."""
        return int(max(1, min(MAX_BATCH, (4 << 30) // (8 * self.N * self.N))))

    def chi2_batch(self, rec, alphas):
        """chi^2 of the regularised solution for B (record, {name: alpha}) pairs -> host array (B,)."""
        rec = np.ascontiguousarray(rec, dtype=np.int32)
        out = np.empty(len(rec))
        mb = self._max_batch()
        for s in range(0, len(rec), mb):
            e = min(len(rec), s + mb)
            B = e - s
            drec, dC, _, _ = self._solve_chunk(rec[s:e], {k: v[s:e] for k, v in alphas.items()}, False, 's_')
            dchi = self._buf('s_chi2', (B,))
            _lib.check(_lib.lib.vi_chi2_f64(self.ctx.handle, B, self.P, self.N, self.At.ptr, dC.ptr, drec.ptr,
                                            self.dW.ptr, self.db.ptr, dchi.ptr), 'vi_chi2_f64')
            tmp = np.empty(B)
            _lib.check(_lib.lib.vi_d2h(self.ctx.handle, tmp.ctypes.data_as(_lib.VOIDP), dchi.ptr, tmp.nbytes), 'd2h')
            out[s:e] = tmp
        return out

    # ------------------------------------------------------------------------------------------
    # ---- warm-started chi^2 evaluation for the Brent phase ------------------------------------------
    def warm_enabled(self):
        if os.environ.get('VINTERP_WARM', '1') == '0':
            return False
        return 8 <= self.N <= 180

    SHARED_WALK_MIN_RECORDS = 8

    def shared_walk_enabled(self):
        # Bracket walk in shared bases (vi_basis_solve_f64): per decade ONE cold decomposition of the reference system
        # (mean weights of the batch) and then 1-4 sweeps per record instead of 8-24.  The reference costs as much as one
        # more record and an extra launch in front of the first walk round, so it needs a few records to pay.
        return (self.warm_enabled() and self.T >= self.SHARED_WALK_MIN_RECORDS and len(self.regularization_list) == 1
                and os.environ.get('VINTERP_SHAREDWALK', '1') != '0')

    def _find_same_below(self, name):
        """Per record: the largest integer k with 10^k below the alpha at which alpha R drops out of AWA + alpha R to
        rounding (vi_reg_floor_f64) - the walk systems of the decades under k are bit-identical to the one at k."""
        T, N = self.T, self.N
        if not T:
            return
        dfl = self._buf('regfloor', (T,))
        _lib.check(_lib.lib.vi_reg_floor_f64(self.ctx.handle, T, N, self.dAWA.ptr, self.R[name].ptr, dfl.ptr),
                   'vi_reg_floor_f64')
        fl = np.empty(T)
        _lib.check(_lib.lib.vi_d2h(self.ctx.handle, fl.ctypes.data_as(_lib.VOIDP), dfl.ptr, fl.nbytes), 'd2h')
        k = np.full(T, -1000.)
        ok = np.isfinite(fl) & (fl > 0) & (fl < 1e300)
        k[ok] = np.ceil(np.log10(fl[ok])) - 1.
        k[ok & ~(np.power(10., np.maximum(k, -300.)) < fl)] -= 1.
        self._same_below[name] = k

    def _warm_reset(self):
        self._warm_slot = {}          # record -> slot of its Brent basis
        self._spec_slot = {}          # (record, bracket midpoint) -> slot of a basis decomposed alongside the walk
        self._last_x = {}             # record -> log10(alpha) of its previous root-finder request
        self._rebased = {}            # record -> times its rotated system has been moved next to the root
        self._nreq = {}               # record -> root-finder requests so far
        self._basis_x = {}            # record -> log10(alpha) its rotated system sits at
        self._basis_slot = {}         # decade -> slot of the reference system's eigenbasis (shared walk)

    REBASE_WITHIN = 3e-2          # decades between two consecutive root-finder requests of a record
    REBASE_AGAIN_AFTER = 20
    REBASE_AGAIN_WITHIN = 1e-5

    def _wants_rebase(self, r, x):
        """Brent's iterates have started to cluster (this request lies within REBASE_WITHIN decades of the record's previous
        one) and the record's rotated system still sits at the middle of the bracket: move it here, once
        (vi_warm_rebase_f64).  The rule looks at the record's own requests only, so what a record sees does not depend on
        the batch it is in."""
        if os.environ.get('VINTERP_REBASE', '1') == '0':
            return False
        last = self._last_x.get(r)
        if last is None:
            return False
        done = self._rebased.get(r, 0)
        sched = self._rebase_schedule()
        if done < len(sched):
            return abs(x - last) < sched[done]
        # once more for the records that are still iterating after REBASE_AGAIN_AFTER requests: they sit on a jump of
        # chi^2 (an eigenvalue of X(alpha) at the cut), Brent bisects down to 2e-12 in 40-60 steps there, and from a basis
        # 1e-3 decades away those solves take 13 sweeps each
        return (done == len(sched) and self._nreq.get(r, 0) >= self.REBASE_AGAIN_AFTER
                and abs(x - last) < self.REBASE_AGAIN_WITHIN and os.environ.get('VINTERP_REBASE2', '1') != '0')

    def _brent_rule(self):
        """h_rebase of vi_brent_warm_f64 / vi_brent_host_one_f64 (10 doubles): the re-basing schedule of the rotated systems and
        the early end on a jump of chi^2 (alpha_search.jump_rule: width in decades, |chi^2 - nu| threshold as a fraction of nu;
        zeros = brentq's own end)."""
        sched = list(self._rebase_schedule()) if os.environ.get('VINTERP_REBASE', '1') != '0' else []
        jr = alpha_search.jump_rule(1.)
        return np.array([len(sched)] + (sched + [0.] * 4)[:4] + [self.REBASE_AGAIN_AFTER, self.REBASE_AGAIN_WITHIN,
                        1. if (sched and os.environ.get('VINTERP_REBASE2', '1') != '0') else 0.,
                        jr[0] if jr else 0., jr[1] if jr else 0.], dtype=np.float64)

    def _rebase_schedule(self):
        """Thresholds (decades between two consecutive requests of a record) at which its rotated system is moved to the
        current request: the k-th move happens the first time two consecutive requests lie closer than schedule[k]."""
        v = getattr(self, '_rebase_sched', None)
        if v is None:
            e = os.environ.get('VINTERP_REBASE_SCHEDULE')
            v = self._rebase_sched = tuple(float(x) for x in e.split(',')) if e else (self.REBASE_WITHIN,)
        return v

    def _warm_buffers(self, tag):
        T, N = self.T, self.N
        return (self._buf(tag + 'V', (T, N, N)), self._buf(tag + 'D1', (T, N, N)), self._buf(tag + 'D2', (T, N, N)),
                self._buf(tag + 'yt', (T, N)))

    def _warm_prepare(self, tag, slots, recs, alphas, name, dC_out, drank_out):
        """Decompose X(alpha) of the given records with eigenvectors into the buffer set `tag`."""
        N, h = self.N, self.ctx.handle
        n = len(recs)
        slot0 = len(slots)
        for k, r in enumerate(recs):
            slots[int(r)] = slot0 + k
        dV, dD1, dD2, dyt = self._warm_buffers(tag)
        dr = self._buf(tag + 'prec', (n,), np.int32).upload(np.asarray(recs, dtype=np.int32))
        da = self._buf(tag + 'palpha', (n,)).upload(np.asarray(alphas, dtype=np.float64))
        _lib.check(_lib.lib.vi_warm_prepare_f64(h, n, N, self.dAWA.ptr, dr.ptr, da.ptr, self.R[name].ptr, self.dy.ptr,
                                                EPS, dC_out, drank_out, dV.offset_ptr(slot0 * N * N),
                                                dD1.offset_ptr(slot0 * N * N), dD2.offset_ptr(slot0 * N * N),
                                                dyt.offset_ptr(slot0 * N)), 'vi_warm_prepare_f64')

    def _warm_solve(self, tag, slots, recs, dalpha_ptr, n, dC_out, drank_out, dsweeps_out=None):
        N, h = self.N, self.ctx.handle
        dV, dD1, dD2, dyt = self._warm_buffers(tag)
        sl = np.array([slots[int(r)] for r in recs], dtype=np.int32)
        dslot = self._buf(tag + 'slot', (n,), np.int32).upload(sl)
        _lib.check(_lib.lib.vi_warm_solve_f64(h, n, N, dD1.ptr, dD2.ptr, dyt.ptr, dV.ptr, dslot.ptr, dalpha_ptr, EPS,
                                              dC_out, drank_out, dsweeps_out), 'vi_warm_solve_f64')

    def chi2_batch_search(self, rec, log10a, name, exact=None):
        """chi^2 requests of the search of `name`; see _chi2_batch_search_raw.  The bracket-walk requests (integer
        log10 alpha) pass through a table first: decades below the record's floor (vi_reg_floor_f64) are one and the same
        system bit for bit and are solved once, and a value once computed is not computed again - the search coroutines
        memoise what they have asked for, but not that -60 and -61 are the same system."""
        rec = np.ascontiguousarray(rec, dtype=np.int32)
        log10a = np.asarray(log10a, dtype=np.float64)
        B = len(rec)
        is_int = log10a == np.floor(log10a)
        if not is_int.any() or os.environ.get('VINTERP_DEDUPE', '1') == '0':
            return self._chi2_batch_search_raw(rec, log10a, name, exact)
        ex = np.zeros(B, dtype=bool) if exact is None else np.asarray(exact, dtype=bool)
        eff = log10a.copy()
        kfl = self._same_below.get(name)
        if kfl is not None:
            low = is_int & (log10a < kfl[rec])
            eff[low] = kfl[rec][low]
        # table of the walk values already known, per (exact?, record, decade); NaN = not yet
        tab = self._walk_cache.get('tab')
        if tab is None or tab.shape[1] != self.T:
            tab = self._walk_cache['tab'] = np.full((2, self.T, 102), np.nan)
        out = np.empty(B)
        ii = np.nonzero(is_int)[0]
        dec = (-eff[ii]).astype(np.int64)
        inside = (dec >= 0) & (dec < 102) & (rec[ii] >= 0) & (rec[ii] < self.T)
        e_i, r_i, d_i = ex[ii].astype(np.int64), rec[ii].astype(np.int64), np.where(inside, dec, 0)
        known = np.where(inside, tab[e_i, np.where(inside, r_i, 0), d_i], np.nan)
        hit_i = ~np.isnan(known)
        out[ii[hit_i]] = known[hit_i]
        # of the misses, one representative per (exact, record, decade)
        miss = ii[~hit_i]
        key = (e_i[~hit_i] * self.T + r_i[~hit_i]) * 102 + d_i[~hit_i]
        key = np.where(inside[~hit_i], key, -1 - np.arange(len(miss)))          # requests outside the table: no sharing
        _, first, inverse = np.unique(key, return_index=True, return_inverse=True)
        reps = miss[first]
        todo = np.sort(np.concatenate([np.nonzero(~is_int)[0], reps]))
        self.stats['walk_same_system'] = self.stats.get('walk_same_system', 0) + B - len(todo)
        if len(todo):
            out[todo] = self._chi2_batch_search_raw(rec[todo], eff[todo], name, ex[todo] if exact is not None else None)
        out[miss] = out[reps][inverse]
        ok = inside[~hit_i]
        tab[e_i[~hit_i][ok], r_i[~hit_i][ok], d_i[~hit_i][ok]] = out[miss][ok]
        return out

    def _chi2_batch_search_raw(self, rec, log10a, name, exact=None):
        """chi^2 for the search of `name` (all other parameters zero), B requests.

        * integer log10(alpha) - the bracket walk - are solved cold for a few records; a batch (>= 8 records) solves them in
          the eigenbases, one per decade, of a reference system built from the batch's mean weights (vi_basis_solve_f64):
          those values only decide signs, the bracket ends Brent starts from come as `exact` requests, which are solved
          cold - a record's answer is the same, bit for bit, whatever batch it is fitted in;
        * root-finder requests (non-integer) are solved in the record's rotated system, which is set up (a cold
          decomposition with eigenvectors) at the middle of the record's unit bracket when its first request arrives."""
        rec = np.ascontiguousarray(rec, dtype=np.int32)
        log10a = np.asarray(log10a, dtype=np.float64)
        B, N = len(rec), self.N
        trace = os.environ.get('VINTERP_TRACE') == '1'
        if trace:
            import time
            t_tr = time.perf_counter()
        is_int = log10a == np.floor(log10a)
        force = getattr(self, '_force_cold', None)
        forced = (np.array([int(r) in force for r in rec.tolist()], dtype=bool) if force else np.zeros(B, dtype=bool))
        if exact is not None:
            forced = forced | np.asarray(exact, dtype=bool)      # reference-grade requests of the search: cold solves
        if (B == 1 and not is_int[0] and not forced[0] and self.warm_enabled() and int(rec[0]) in self._warm_slot
                and not self._wants_rebase(int(rec[0]), float(log10a[0]))):
            self._last_x[int(rec[0])] = float(log10a[0])
            self._nreq[int(rec[0])] = self._nreq.get(int(rec[0]), 0) + 1
            # a single root-finder iterate of a record whose rotated system exists: one library call, no uploads
            dV, dD1, dD2, dyt = self._warm_buffers('w_')
            scratch = self._buf('w_one', (N + 8,))
            back = np.zeros(3)                  # chi^2, an internal word, the sweep count (low 32 bits)
            r = int(rec[0])
            _lib.check(_lib.lib.vi_warm_chi2_one_f64(self.ctx.handle, N, self.P, dD1.ptr, dD2.ptr, dyt.ptr, dV.ptr,
                                                     self._warm_slot[r], self._exp10(log10a[0]), EPS,
                                                     self.At.ptr, r, self.dW.ptr, self.db.ptr, scratch.ptr,
                                                     back.ctypes.data_as(C.POINTER(C.c_double))), 'vi_warm_chi2_one_f64')
            self.stats['solves'] += 1
            self.stats['launches'] += 1
            self.stats['warm_solves'] = self.stats.get('warm_solves', 0) + 1
            if int(back[2:3].view(np.int32)[0]) > self.max_sweeps():
                # the sweep cap ended the solve: its value decides nothing - the same system from X(alpha) itself
                self.stats['unconverged_resolved'] = self.stats.get('unconverged_resolved', 0) + 1
                return self._cold_chi2(rec, np.power(10., log10a), name)
            if trace:
                print('[search round] B=1 warm (single call)  %.2f ms  log10a[0]=%.12f' %
                      ((time.perf_counter() - t_tr) * 1e3, log10a[0]))
            return np.array([back[0]])
        if not self.warm_enabled():
            al = {n: (np.power(10., log10a) if n == name else np.zeros(B)) for n in self.regularization_list}
            return self.chi2_batch(rec, al)
        alpha = np.power(10., log10a)
        if not is_int.all():
            # the root finder's abscissae: 10^x by the routine the device-side iteration uses (csrc/vi_exp10.h), so that
            # host-driven and device-side iterations form the same alpha from the same x
            ni = np.ascontiguousarray(log10a[~is_int])
            out_ = np.empty_like(ni)
            _lib.check(_lib.lib.vi_exp10_f64(ni.ctypes.data_as(_lib.VOIDP), out_.ctypes.data_as(_lib.VOIDP), len(ni)), 'vi_exp10_f64')
            alpha[~is_int] = out_
        if (self.T == 1 and B >= 8 and is_int.all() and not forced.any() and not self._warm_slot and not self._spec_slot
                and os.environ.get('VINTERP_SPECULATE', '1') != '0'):
            return self._walk_with_speculative_bases(rec, log10a, alpha, name, trace)
        shared = (is_int & ~forced if (self._ref_rec is not None and self.shared_walk_enabled())
                  else np.zeros(B, dtype=bool))
        # root-finder requests (non-integer): a record without a rotated system yet gets it from ONE of its
        # requests - the middle one when a multisection round asks for many, so the basis is nearest to all
        warm = np.zeros(B, dtype=bool)
        rebase = np.zeros(B, dtype=bool)
        by_rec = {}
        for j in np.nonzero(~is_int & ~forced)[0].tolist():
            by_rec.setdefault(int(rec[j]), []).append(j)
        need = {}
        for r, js in by_rec.items():
            warm[js] = True
            if len(js) == 1:
                x = float(log10a[js[0]])
                if r in self._warm_slot and self._wants_rebase(r, x):
                    warm[js] = False
                    rebase[js] = True
                    self._rebased[r] = self._rebased.get(r, 0) + 1
                    self._basis_x[r] = x
                self._last_x[r] = x
                self._nreq[r] = self._nreq.get(r, 0) + 1
            if r not in self._warm_slot:
                # The rotated system of a record is set up at the MIDDLE of its unit bracket, 10^(floor(x) + 1/2),
                # whatever the request that triggers it (Brent's first iterate, a multisection sample): the warm chi^2 is
                # then one function of alpha per record, independent of the batch the record is fitted in and of the
                # root finder's path - alone or among 999 others, a record sees the same values and Brent takes the
                # same steps.  (It used to be set up at the first request; a record fitted alone, whose first request
                # is a multisection sample, then got another basis, other rounding and sometimes another of the several
                # roots of a default-order bracket than the same record inside a batch.)
                need[r] = math.floor(float(log10a[js[0]])) + 0.5
                self._basis_x[r] = need[r]
                if (r, need[r]) in self._spec_slot:      # decomposed alongside the walk (_walk_with_speculative_bases)
                    self._finish_speculative(r, self._spec_slot[(r, need[r])])
                    del need[r]
        if need:
            recs_n = sorted(need)
            scratchC = self._buf('wp_scratchC', (len(recs_n), N))
            scratchR = self._buf('wp_scratchR', (len(recs_n),), np.int32)
            self._warm_prepare('w_', self._warm_slot, recs_n, [float(np.power(10., need[r])) for r in recs_n], name,
                               scratchC.ptr, scratchR.ptr)
        cold = (is_int & ~shared) | forced               # forced: records whose search is being redone cold
        sh_idx = np.nonzero(shared)[0]
        sh_idx = sh_idx[np.argsort(log10a[sh_idx], kind='stable')]          # by decade: one basis after the other
        # warm solves cost more the further alpha is from where the record's rotated system sits: longest first, so
        # that a launch does not end on one straggler started last
        w_idx = np.nonzero(warm)[0]
        if len(w_idx) > 256 and os.environ.get('VINTERP_LPT', '1') != '0':
            dist = np.abs(log10a[w_idx] - np.array([self._basis_x.get(int(r), 0.) for r in rec[w_idx].tolist()]))
            w_idx = w_idx[np.argsort(-dist, kind='stable')]
        order = np.concatenate([np.nonzero(cold)[0], sh_idx, w_idx, np.nonzero(rebase)[0]])
        nc, nsh, nw, nrb = int(cold.sum()), len(sh_idx), int(warm.sum()), int(rebase.sum())
        h = self.ctx.handle
        dCall = self._buf('w_C', (B, N))
        drank = self._buf('w_rank', (B,), np.int32)
        dsw = self._buf('w_sweeps', (B,), np.int32)
        rec_o, alpha_o = rec[order], alpha[order]
        drec = self._buf('w_rec', (B,), np.int32).upload(rec_o)
        dal = self._buf('w_alpha', (B,)).upload(alpha_o)
        if nc:
            mb = self._max_batch()
            dX = self._buf('w_X', (min(nc, mb), N, N))
            for s0 in range(0, nc, mb):
                bc = min(mb, nc - s0)
                _lib.check(_lib.lib.vi_form_system_f64(h, bc, N, self.dAWA.ptr, drec.offset_ptr(s0), dal.offset_ptr(s0),
                                                       self.R[name].ptr, dX.ptr), 'vi_form_system_f64')
                _lib.check(_lib.lib.vi_solve_trunc_f64(h, bc, N, dX.ptr, self.dy.ptr, drec.offset_ptr(s0), EPS,
                                                       dCall.offset_ptr(s0 * N), drank.offset_ptr(s0), N * EPS, None),
                           'vi_solve_trunc_f64')
        o = nc
        if nsh:
            decades = np.rint(log10a[sh_idx]).astype(np.int64)
            new_k = sorted(set(decades.tolist()) - set(self._basis_slot), reverse=True)
            if new_k and not self._basis_slot:
                # the reference bases of ALL the decades the walk can ask for, in one launch: decade by decade as the walk
                # proceeds, every round waited for its own cold decompositions (3.5 ms + eigenvectors, seven times)
                kfl = self._same_below.get(name)
                lowest = int(max(-101, np.min(kfl))) if kfl is not None and len(kfl) else -101
                new_k = sorted(set(new_k) | set(range(0, min(lowest, min(new_k)) - 1, -1)), reverse=True)
            dV, dD1, dD2, dyt = (self._buf('sb_V', (102, N, N)), self._buf('sb_D1', (102, N, N)),
                                 self._buf('sb_D2', (102, N, N)), self._buf('sb_yt', (102, N)))
            if new_k:
                if len(self._basis_slot) + len(new_k) > 102:
                    raise RuntimeError('shared walk: more than 102 decades requested')
                slot0 = len(self._basis_slot)
                for i, k in enumerate(new_k):
                    self._basis_slot[k] = slot0 + i
                n = len(new_k)
                dr = self._buf('sb_prec', (n,), np.int32).upload(np.full(n, self._ref_rec, dtype=np.int32))
                da = self._buf('sb_palpha', (n,)).upload(np.power(10., np.asarray(new_k, dtype=np.float64)))
                scratchC = self._buf('sb_scratchC', (n, N))
                scratchR = self._buf('sb_scratchR', (n,), np.int32)
                _lib.check(_lib.lib.vi_warm_prepare_f64(h, n, N, self.dAWA.ptr, dr.ptr, da.ptr, self.R[name].ptr,
                                                        self.dy.ptr, EPS, scratchC.ptr, scratchR.ptr,
                                                        dV.offset_ptr(slot0 * N * N), dD1.offset_ptr(slot0 * N * N),
                                                        dD2.offset_ptr(slot0 * N * N), dyt.offset_ptr(slot0 * N)),
                           'vi_warm_prepare_f64')
                self.stats['solves'] += n
                self.stats['reference_solves'] = self.stats.get('reference_solves', 0) + n
            dbs = self._buf('sb_slot', (nsh,), np.int32).upload(
                np.array([self._basis_slot[k] for k in decades.tolist()], dtype=np.int32))
            _lib.check(_lib.lib.vi_basis_solve_f64(h, nsh, N, self.dAWA.ptr, self.dy.ptr, drec.offset_ptr(o),
                                                   dbs.ptr, dal.offset_ptr(o), dV.ptr, dD2.ptr, EPS,
                                                   dCall.offset_ptr(o * N), drank.offset_ptr(o), dsw.offset_ptr(o)),
                       'vi_basis_solve_f64')
            o += nsh
        if nw and not nrb:
            self._warm_solve('w_', self._warm_slot, rec_o[o:o + nw], dal.offset_ptr(o), nw, dCall.offset_ptr(o * N),
                             drank.offset_ptr(o), dsw.offset_ptr(o))
        elif nrb:
            # the plain warm solves of the round ride in the launch of the re-basing ones (a launch lasts as long as its
            # slowest system, however few it holds)
            n = nw + nrb
            dV, dD1, dD2, dyt = self._warm_buffers('w_')
            sl = np.array([self._warm_slot[int(r)] for r in rec_o[o:o + n]], dtype=np.int32)
            dslot = self._buf('w_rbslot', (n,), np.int32).upload(sl)
            _lib.check(_lib.lib.vi_warm_rebase_f64(h, n, nw, N, self.dAWA.ptr, self.R[name].ptr, self.dy.ptr, drec.offset_ptr(o),
                                                   dslot.ptr, dal.offset_ptr(o), EPS, dV.ptr, dD1.ptr, dD2.ptr, dyt.ptr,
                                                   dCall.offset_ptr(o * N), drank.offset_ptr(o), dsw.offset_ptr(o)),
                       'vi_warm_rebase_f64')
            self.stats['rebased'] = self.stats.get('rebased', 0) + nrb
        dchi = self._buf('w_chi2', (B,))
        _lib.check(_lib.lib.vi_chi2_f64(h, B, self.P, N, self.At.ptr, dCall.ptr, drec.ptr, self.dW.ptr, self.db.ptr,
                                        dchi.ptr), 'vi_chi2_f64')
        tmp = np.empty(B)
        _lib.check(_lib.lib.vi_d2h(h, tmp.ctypes.data_as(_lib.VOIDP), dchi.ptr, tmp.nbytes), 'd2h')
        if B > nc:
            # solves in a rotated system (shared walk bases, warm iterates) that the sweep cap ended before they converged:
            # their chi^2 must not decide a sign or steer Brent - the same systems again from X(alpha) itself.  How many
            # sweeps a rotated system takes depends on how far the record lies from the basis it is solved in (a record
            # with most of its points dropped sits far from the batch's mean system).
            sw = np.empty(B - nc, dtype=np.int32)
            _lib.check(_lib.lib.vi_d2h(h, sw.ctypes.data_as(_lib.VOIDP), dsw.offset_ptr(nc), sw.nbytes), 'd2h')
            bad = nc + np.nonzero(sw > self.max_sweeps())[0]
            if len(bad):
                tmp[bad] = self._cold_chi2(rec_o[bad], alpha_o[bad], name)
                self.stats['unconverged_resolved'] = self.stats.get('unconverged_resolved', 0) + len(bad)
        out = np.empty(B)
        out[order] = tmp
        if trace:
            print('[search round] B=%d cold=%d shared=%d warm=%d rebase=%d  %.2f ms  log10a[0]=%.12f' %
                  (B, nc, nsh, nw, nrb, (time.perf_counter() - t_tr) * 1e3, log10a[0]))
        self.stats['solves'] += B
        self.stats['launches'] += 1
        self.stats['warm_solves'] = self.stats.get('warm_solves', 0) + nw + nrb
        self.stats['shared_solves'] = self.stats.get('shared_solves', 0) + nsh
        return out

    DEVICE_BRENT_MIN_RECORDS = 8

    def device_brent_enabled(self):
        """Brent's iteration of the whole batch in one launch (vi_brent_warm_f64, csrc/vi_brent.hip) instead of one round of
        launches per iterate: a workgroup owns a record from its bracket to its root, re-basing its rotated system as the host
        path does.  Bit for bit the host-driven iteration (tests/test_gpu_search_stages.py), so a record's answer does not
        depend on which of the two served it.  From eight records on: a record fitted alone keeps the host-driven rounds
        (one workgroup would do the products of the re-basing alone that the host path spreads over the chip).
        VINTERP_DEVICE_BRENT=0 / 1 forces either.  Measured, 1000 records: 543 -> 466 ms in one pipeline, 481 -> 419 in four."""
        e = os.environ.get('VINTERP_DEVICE_BRENT', 'auto')
        if e == '0' or not self.warm_enabled() or len(self.regularization_list) != 1:
            return False
        if len(self._rebase_schedule()) > 4:
            return False
        if not _lib.lib.vi_brent_warm_supported(self.N, self.P):
            # records with so many data points that their chi^2 partial sums do not fit beside the system in a CU's LDS: the
            # host-driven rounds, which have no limit (the reference accepts any P)
            return False
        return e == '1' or self.T >= self.DEVICE_BRENT_MIN_RECORDS

    def _device_brent(self, recs, brackets, name):
        """Brent's iteration for the records `recs` (brackets: dicts with alpha, alpha0, val, val0, nu) on the device.
        Rotated systems are set up at the middle of each record's unit bracket, as the host path does; returns one
        (root, iterations, funcalls, other_end) per record, None where the kernel gave up (a solve ended by the sweep cap)."""
        N, h, n = self.N, self.ctx.handle, len(recs)
        recs = [int(r) for r in recs]
        need = [r for r in recs if r not in self._warm_slot]
        if need:
            mids = {r: math.floor(min(b['alpha'], b['alpha0'])) + 0.5 for r, b in zip(recs, brackets)}
            scratchC = self._buf('wp_scratchC', (len(need), N))
            scratchR = self._buf('wp_scratchR', (len(need),), np.int32)
            for r in need:
                self._basis_x[r] = mids[r]
            self._warm_prepare('w_', self._warm_slot, need, [float(np.power(10., mids[r])) for r in need], name,
                               scratchC.ptr, scratchR.ptr)
            self.stats['solves'] += len(need)
        dV, dD1, dD2, dyt = self._warm_buffers('w_')
        up = lambda key, a, dt: self._buf('db_' + key, (n,), dt).upload(np.asarray(a, dtype=dt))      # noqa: E731
        drec, dslot = up('rec', recs, np.int32), up('slot', [self._warm_slot[r] for r in recs], np.int32)
        dxa, dxb = up('xa', [b['alpha'] for b in brackets], np.float64), up('xb', [b['alpha0'] for b in brackets], np.float64)
        dfa, dfb = up('fa', [b['val'] for b in brackets], np.float64), up('fb', [b['val0'] for b in brackets], np.float64)
        dnu = up('nu', [b['nu'] for b in brackets], np.float64)
        droot, dother = self._buf('db_root', (n,)), self._buf('db_other', (n,))
        dit, dfc, dst = (self._buf('db_' + k, (n,), np.int32) for k in ('it', 'fc', 'st'))
        rule = self._brent_rule()
        _lib.check(_lib.lib.vi_brent_warm_f64(h, n, N, self.P, dD1.ptr, dD2.ptr, dyt.ptr, dV.ptr, self.dAWA.ptr,
                                              self.R[name].ptr, self.dy.ptr, rule.ctypes.data_as(_lib.VOIDP), self.At.ptr,
                                              self.dW.ptr, self.db.ptr, drec.ptr, dslot.ptr, dxa.ptr, dxb.ptr, dfa.ptr, dfb.ptr, dnu.ptr,
                                              EPS, droot.ptr, dother.ptr, dit.ptr, dfc.ptr, dst.ptr), 'vi_brent_warm_f64')

        def down(d, dt):
            a = np.empty(n, dtype=dt)
            _lib.check(_lib.lib.vi_d2h(h, a.ctypes.data_as(_lib.VOIDP), d.ptr, a.nbytes), 'd2h')
            return a
        root, other, its, fcs, sts = down(droot, np.float64), down(dother, np.float64), down(dit, np.int32), \
            down(dfc, np.int32), down(dst, np.int32)
        self.stats['rebased'] = self.stats.get('rebased', 0) + int((sts >> 8).sum())
        sts = sts & 0xff
        if np.any(sts == 3):
            raise RuntimeError('Failed to converge after %d iterations.' % alpha_search.MAXITER)
        self.stats['solves'] += int(fcs.sum())
        self.stats['launches'] += 1
        self.stats['warm_solves'] = self.stats.get('warm_solves', 0) + int(fcs.sum())
        self.stats['device_brent_records'] = self.stats.get('device_brent_records', 0) + int((sts == 0).sum())
        if np.any(sts == 2):
            self.stats['unconverged_resolved'] = self.stats.get('unconverged_resolved', 0) + int((sts == 2).sum())
        return [(float(root[j]), int(its[j]), int(fcs[j]), float(other[j])) if sts[j] == 0 else None for j in range(n)]

    def host_loop_brent_enabled(self):
        """Brent's iteration of a record fitted ALONE as one library call (vi_brent_host_one_f64: the host-driven loop in C,
        the state machine of the device kernel compiled for the host) instead of one call per function value with the search
        coroutine in between: same requests, same values, same bits.  VINTERP_HOST_LOOP_BRENT=0 keeps the loop in Python."""
        if os.environ.get('VINTERP_HOST_LOOP_BRENT', '1') == '0' or not self.warm_enabled() or len(self.regularization_list) != 1:
            return False
        return self.T == 1 and len(self._rebase_schedule()) <= 4

    def _host_loop_brent(self, recs, brackets, name):
        """_device_brent's contract for the records of a fit that is driven from the host (one record)."""
        N, h = self.N, self.ctx.handle
        out = []
        rule = self._brent_rule()
        for r, b in zip([int(r) for r in recs], brackets):
            if r not in self._warm_slot:
                # the rotated system at the middle of the record's unit bracket: decomposed alongside the walk, or now
                mid = math.floor(min(b['alpha'], b['alpha0'])) + 0.5
                self._basis_x[r] = mid
                if (r, mid) in self._spec_slot:
                    self._finish_speculative(r, self._spec_slot[(r, mid)])
                else:
                    scratchC = self._buf('wp_scratchC', (1, N))
                    scratchR = self._buf('wp_scratchR', (1,), np.int32)
                    self._warm_prepare('w_', self._warm_slot, [r], [float(np.power(10., mid))], name, scratchC.ptr, scratchR.ptr)
                    self.stats['solves'] += 1
            dV, dD1, dD2, dyt = self._warm_buffers('w_')
            scratch = self._buf('w_one', (N + 8,))
            res = np.zeros(6)
            _lib.check(_lib.lib.vi_brent_host_one_f64(h, N, self.P, dD1.ptr, dD2.ptr, dyt.ptr, dV.ptr, self.dAWA.ptr,
                                                      self.R[name].ptr, self.dy.ptr, rule.ctypes.data_as(_lib.VOIDP), self.At.ptr,
                                                      self.dW.ptr, self.db.ptr, r, self._warm_slot[r], float(b['alpha']),
                                                      float(b['alpha0']), float(b['val']), float(b['val0']), float(b['nu']), EPS,
                                                      scratch.ptr, res.ctypes.data_as(_lib.VOIDP)), 'vi_brent_host_one_f64')
            st, fc = int(res[4]), int(res[3])
            if st == 3:
                raise RuntimeError('Failed to converge after %d iterations.' % alpha_search.MAXITER)
            self.stats['solves'] += fc
            self.stats['launches'] += fc
            self.stats['warm_solves'] = self.stats.get('warm_solves', 0) + fc
            self.stats['rebased'] = self.stats.get('rebased', 0) + int(res[5])
            if st == 2:
                self.stats['unconverged_resolved'] = self.stats.get('unconverged_resolved', 0) + 1
                out.append(None)
            else:
                out.append((float(res[0]), int(res[2]), fc, float(res[1])))
        return out

    @staticmethod
    def _exp10(x):
        """10^x for one root-finder abscissa by the library's routine (see _chi2_batch_search_raw)."""
        a, o = np.array([float(x)]), np.empty(1)
        _lib.check(_lib.lib.vi_exp10_f64(a.ctypes.data_as(_lib.VOIDP), o.ctypes.data_as(_lib.VOIDP), 1), 'vi_exp10_f64')
        return float(o[0])

    def max_sweeps(self):
        v = getattr(self, '_max_sweeps', None)
        if v is None:
            v = self._max_sweeps = int(_lib.lib.vi_max_sweeps())
        return v

    def _cold_chi2(self, rec, alpha, name):
        """chi^2 of (record, alpha) pairs from cold solves of X(alpha) itself (other parameters zero)."""
        rec = np.ascontiguousarray(rec, dtype=np.int32)
        al = {n: (np.asarray(alpha, dtype=np.float64) if n == name else np.zeros(len(rec))) for n in self.regularization_list}
        return self.chi2_batch(rec, al)

    def _walk_with_speculative_bases(self, rec, log10a, alpha, name, trace):
        """The bracket walk of a record fitted ALONE, with the rotated systems of all its candidate brackets.

        A single record's fit is a chain of dependent launches on an otherwise empty GPU, and a launch of up to 256 systems
        lasts as long as its slowest one.  The walk (one launch: every decade at once) is followed by the decomposition, with
        eigenvectors, of X at the middle of the bracket it finds - another cold solve, another 3.5 ms of the ~28.  Which
        bracket that will be is not known before the walk is done, but there are at most 101 candidates and the launch
        has room: the walk systems and the midpoint systems of all brackets above the record's floor are decomposed in
        ONE launch, and the basis Brent needs is there when the walk ends.  Same kernels on the same systems: the walk
        values and the chosen basis are bit for bit what the two separate launches gave."""
        N, h, B = self.N, self.ctx.handle, len(rec)
        kfl = self._same_below.get(name)
        mids = []
        for r, k in sorted(set(zip(rec.tolist(), log10a.tolist()))):
            lo = k - 1.
            if lo >= -101. and (kfl is None or lo >= kfl[r]):
                mids.append((int(r), k - 0.5))
        n = B + len(mids)
        recs = np.concatenate([rec, np.array([r for r, _ in mids], dtype=np.int32)]).astype(np.int32)
        als = np.concatenate([alpha, np.power(10., np.array([x for _, x in mids], dtype=np.float64))])
        logb = int(_lib.lib.vi_rotation_log_bytes(N))
        dlog = self._buf('sp_log', (n * (logb // 8),))
        dnr = self._buf('sp_nround', (n,), np.int32)
        dr = self._buf('sp_rec', (n,), np.int32).upload(recs)
        da = self._buf('sp_alpha', (n,)).upload(als)
        dC = self._buf('sp_C', (n, N))
        drk = self._buf('sp_rank', (n,), np.int32)
        _lib.check(_lib.lib.vi_decompose_f64(h, n, N, self.dAWA.ptr, dr.ptr, da.ptr, self.R[name].ptr, self.dy.ptr, EPS,
                                             dC.ptr, drk.ptr, dlog.ptr, dnr.ptr), 'vi_decompose_f64')
        for j, key in enumerate(mids):
            self._spec_slot[key] = B + j          # index of the system's rotation log
        self._spec_name = name
        dchi = self._buf('sp_chi2', (B,))
        _lib.check(_lib.lib.vi_chi2_f64(h, B, self.P, N, self.At.ptr, dC.ptr, dr.ptr, self.dW.ptr, self.db.ptr, dchi.ptr),
                   'vi_chi2_f64')
        out = np.empty(B)
        _lib.check(_lib.lib.vi_d2h(h, out.ctypes.data_as(_lib.VOIDP), dchi.ptr, out.nbytes), 'd2h')
        self.stats['solves'] += n
        self.stats['launches'] += 1
        self.stats['speculative_bases'] = self.stats.get('speculative_bases', 0) + len(mids)
        if trace:
            print('[search round] B=%d walk + %d speculative bracket bases in one launch' % (B, len(mids)))
        return out

    def _finish_speculative(self, r, j):
        """Eigenvectors and rotated system of record r from rotation log j of the walk launch -> the record's slot."""
        N, h = self.N, self.ctx.handle
        logb = int(_lib.lib.vi_rotation_log_bytes(N))
        slot = len(self._warm_slot)
        self._warm_slot[r] = slot
        dV, dD1, dD2, dyt = self._warm_buffers('w_')
        owner = self._scratch_of if self._scratch_of is not None else self
        dlog, dnr = owner._bufs['sp_log'], owner._bufs['sp_nround']
        dr = self._buf('sp_rec1', (1,), np.int32).upload(np.array([r], dtype=np.int32))
        _lib.check(_lib.lib.vi_warm_finish_f64(h, 1, N, dlog.offset_ptr(j * (logb // 8)), dnr.offset_ptr(j), self.dAWA.ptr,
                                               dr.ptr, self.R[self._spec_name].ptr, self.dy.ptr, dV.offset_ptr(slot * N * N),
                                               dD1.offset_ptr(slot * N * N), dD2.offset_ptr(slot * N * N),
                                               dyt.offset_ptr(slot * N)), 'vi_warm_finish_f64')

    def default_prefetch(self):
        # walk prefetch: the whole alpha = 0 .. -101 table in one launch for a single record (latency-bound), a few steps
        # ahead for large batches (the launch is already full; don't waste solves).  Also beyond the in-LDS solver
        # (N > 180, rocSOLVER syevd): its batched form costs 10.6 ms per system in launches of 4 but 2.7 ms in launches of
        # 64 or more - one syevd at N = 1152 is ~900 small dependent kernels, and only a batch fills the GPU - so a
        # single N = 1152 record takes 481 ms with the whole walk in one launch against 1164 ms four values at a time
        # (measured; running several syevd calls from concurrent host threads instead made it slower, 1460 ms).
        if os.environ.get('VINTERP_PREFETCH'):
            return int(os.environ['VINTERP_PREFETCH'])
        if self._ref_rec is not None and self.shared_walk_enabled():
            # in the shared bases a walk system costs a tenth of a cold one: fewer, fuller rounds (measured: 100 records
            # 224 -> 214 ms, 300 records in four pipelines 327 -> 323 ms, 1000 records unchanged)
            return int(max(8, min(102, 32768 // max(1, self.T))))          # (1000 records: 8 -> 32 decades per round, 472 -> 450 ms)
        return int(max(8, min(102, 2048 // max(1, self.T))))

    def default_multisection(self):
        mode = os.environ.get('VINTERP_ROOT', 'auto')
        if mode == 'brent':
            return 0
        # chi^2(alpha) - nu is not monotone (the curvature matrix is indefinite) and can cross zero several times
        # inside one unit bracket; Brent's iterate sequence decides which root the reference returns (measured on
        # the screened MAXK=8, MAXL=2 fixture: roots at -28.4698 and -28.0745 in [-29,-28]).  The multisection of
        # alpha_search.multisection_gen is therefore GUARDED: it proceeds only while each round's dense sampling
        # shows exactly one sign change and otherwise hands the bracket back to the exact Brent iteration.
        # It pays when the GPU is otherwise idle (few records: ~15 dependent solves become 3 batched rounds); a
        # refused round costs one launch (measured at N = 144, where the indefinite curvature matrix puts poles of
        # chi^2 inside most brackets and the guard refuses: T = 1  38.0 vs 39.4 ms, T = 2  84 vs 94, T = 8  121 vs
        # 117, T = 16  153 vs 145 - the refused round also sets up the rotated system Brent then uses).
        # Only where the in-LDS solver serves the rounds: at orders beyond it (N > 180, rocSOLVER, ~0.15 s per solve
        # at N = 1152) K extra solves per round are far dearer than Brent's dependent ones.
        # Not at the default order and beyond (N >= 100): there the guard refuses on most brackets and the attempt is pure
        # cost - one launch of 254 warm systems, 3.1 ms of a 28.6 ms single-record fit (bench.py, 25.5 ms without it; the
        # rotated system sits at the middle of the bracket either way now, so Brent's steps and the answer are the same).
        # The number of samples per round does not depend on the number of records (it used to be 256 / T - 1, so a record's
        # last digits depended on whether it was fitted alone or with one to three others): 63 samples, 64-fold shrinkage,
        # four rounds to 1e-7.  Batches of five records or more go straight to Brent, whose root agrees with the
        # multisection's to ~1e-7 decades (tests/test_gpu_fit.py::test_small_order_root_does_not_depend_on_the_batch).
        if mode == 'multisection' or (self.T <= 4 and self.warm_enabled() and self.N < 100):
            return 63
        return 0

    def search(self, npts, prefetch=None, multisection=None, only=None, cold=False):
        """find_reg_param with method 'chi2' for every loaded record (interpolate.py:97-147).

        npts[t] = number of finite points of record t, or None to skip it.  Returns a list of
        {name: alpha} dicts (NaN where the search fails) and per-name search info.
        only: restrict to these records (the others are reported as skipped); cold: serve every chi^2 request of
        the search from cold solves of the untransformed system (no rotated-system warm start) - the path the
        consistency guard of fit_resident() falls back to."""
        T = self.T
        if prefetch is None:
            prefetch = self.default_prefetch()
        if multisection is None:
            multisection = self.default_multisection()
        if only is not None:
            only = set(int(t) for t in only)
            npts = [n if t in only else None for t, n in enumerate(npts)]
        self._force_cold = set(range(T)) if (cold and only is None) else (set(only) if cold else set())
        if cold:
            multisection = 0                    # the reference's own iteration only
        params = [dict() for _ in range(T)]
        infos = {}
        for name in self.regularization_list:
            self._warm_reset()
            self._find_same_below(name)
            self._walk_cache = {}

            def evaluate(rec, log10a, exact=None, _name=name):
                return self.chi2_batch_search(rec, log10a, _name, exact)
            # walk values from the shared bases only decide signs; the bracket ends Brent starts from are asked for
            # again from cold solves (alpha_search.chi2_search_gen, refine)
            refine = bool(self._ref_rec is not None and self.shared_walk_enabled() and not cold)
            # Brent's iteration on arrays pays from a dozen records on (NumPy's per-call overhead on arrays of one or two
            # elements is several times the coroutine's step)
            solver = None
            if self.device_brent_enabled() and not cold and not multisection:
                def solver(recs, brs, _name=name):
                    return self._device_brent(recs, brs, _name)
            elif self.host_loop_brent_enabled() and not cold and not multisection:
                def solver(recs, brs, _name=name):
                    return self._host_loop_brent(recs, brs, _name)
            if not multisection and T >= 16 and os.environ.get('VINTERP_TABLE_WALK', '1') != '0':
                # the walks of the whole batch on one (records x decades) table: the coroutines' requests and decisions
                # without the coroutines (110 ms of interpreter per 1000 records, which concurrent pipelines cannot share)
                alphas, outcomes, info, nev = alpha_search.run_table_batched(npts, evaluate, prefetch=prefetch, refine=refine,
                                                                             brent_solver=solver)
            else:
                alphas, outcomes, info, nev = alpha_search.run_batched(npts, evaluate, prefetch=prefetch,
                                                                       multisection=multisection, refine=refine,
                                                                       vector_brent=T >= 16, brent_solver=solver)
            for t in range(T):
                params[t][name] = alphas[t]
            infos[name] = dict(outcomes=outcomes, info=info, evaluations=nev)
        self._force_cold = set()
        return params, infos

    # ---- generalised cross validation (interpolate.py:263-351) ------------------------------------------
    def gcv_objective(self, t, log10a, name, pidx):
        """Interpolate.gcvobjfunct for record t: sum over the listed points of the leave-one-out residuals."""
        pidx = np.ascontiguousarray(pidx, dtype=np.int32)
        npnt, N, P = len(pidx), self.N, self.P
        dp = self._buf('g_pidx', (npnt,), np.int32).upload(pidx)
        dres = self._buf('g_res', (npnt,))
        alpha = float(np.power(10., float(np.squeeze(log10a))))
        _lib.check(_lib.lib.vi_gcv_terms_f64(self.ctx.handle, npnt, P, N, self.At.ptr, dp.ptr,
                                             self.dAWA.offset_ptr(t * N * N), self.dy.offset_ptr(t * N),
                                             self.dW.offset_ptr(t * P), self.db.offset_ptr(t * P), alpha,
                                             self.R[name].ptr, EPS, dres.ptr), 'vi_gcv_terms_f64')
        res = np.empty(npnt)
        _lib.check(_lib.lib.vi_d2h(self.ctx.handle, res.ctypes.data_as(_lib.VOIDP), dres.ptr, res.nbytes), 'd2h')
        self.stats['solves'] += npnt
        self.stats['launches'] += 1
        return sum(res.tolist())                    # the reference's left-to-right Python sum

    def search_gcv(self, point_lists):
        """find_reg_param with method 'gcv' for every loaded record: Nelder-Mead from log10(alpha) = -20
        (scipy.optimize.minimize, exactly the reference's call at interpolate.py:291) on the device objective.
        point_lists[t]: indices of the finite data points of record t, or None to skip (NaN)."""
        import scipy.optimize
        params = [dict() for _ in range(self.T)]
        infos = {}
        for name in self.regularization_list:
            outcomes = []
            for t in range(self.T):
                if point_lists[t] is None:
                    params[t][name] = float('nan')
                    outcomes.append('skipped')
                    continue
                sol = scipy.optimize.minimize(lambda a, _t=t: self.gcv_objective(_t, a, name, point_lists[_t]), -20.,
                                              method='Nelder-Mead')
                if sol.success:
                    params[t][name] = float(np.power(10., sol.x[0]))
                    outcomes.append('minimum')
                else:                               # ValueError('Minima of GCV function could not be found') -> NaN
                    params[t][name] = float('nan')
                    outcomes.append('no_minimum')
            infos[name] = dict(outcomes=outcomes)
        return params, infos

    def finalize(self, params, calccov=True, only=None, out=None, compact=False, defer_cov=False):
        """Final eval_C(calccov=True) + chi^2 for every record (interpolate.py:566-569).

        Records whose parameters contain NaN become NaN rows (interpolate.py:558-563); with `only`, so do the
        records not listed.  compact: arrays with one row per solved record only, and the list of those records as a fifth
        result (the guard re-finalises a handful of records of a batch: no batch-sized arrays for them).
        The result arrays are written once: rows of solved records by the download itself (straight into the rows when the
        chunk's records are consecutive - 166 MB of covariances per 1000 records used to pass through a staging array and
        a row-by-row copy after a NaN fill of the whole array), NaN only into the rows of the others.
        defer_cov: the covariances stay on the device (a buffer of their own) and come down beside the stream in a host
        thread (vi_d2h_side) while the caller goes on launching - self._cov_pending is that thread, to be joined (join_cov)
        before the covariance array is read or written."""
        T, N = self.T, self.N
        good = [t for t in range(T) if (only is None or t in only)
                and not np.any(np.isnan([params[t][n] for n in self.regularization_list]))]
        if compact:
            R = len(good)
            Coeffs, Cov = np.empty((R, N)), (np.empty((R, N, N)) if calccov else None)
            chi, ranks = np.empty(R), np.empty(R, dtype=np.int32)
            rows = np.arange(R, dtype=np.int64)
        else:
            if out is not None:                     # caller's arrays (a pipeline's share of the batch's result)
                Coeffs, Cov, chi, ranks = out
            else:
                Coeffs = np.empty((T, N))
                Cov = np.empty((T, N, N)) if calccov else None
                chi = np.empty(T)
                ranks = np.empty(T, dtype=np.int32)
            rows = np.asarray(good, dtype=np.int64)
            miss = np.ones(T, dtype=bool)
            miss[rows] = False
            if miss.any():
                Coeffs[miss] = np.nan
                chi[miss] = np.nan
                ranks[miss] = -1
                if Cov is not None:
                    Cov[miss] = np.nan
        h = self.ctx.handle

        def down(dst, dev, r0, B, shape, dtype):
            """B rows of `dev` into dst[rows[r0 : r0 + B]]"""
            rr = rows[r0:r0 + B]
            if B and int(rr[-1] - rr[0]) == B - 1 and dst.flags['C_CONTIGUOUS']:
                view = dst[int(rr[0]):int(rr[0]) + B]                  # consecutive rows: the download lands in place
                _lib.check(_lib.lib.vi_d2h(h, view.ctypes.data_as(_lib.VOIDP), dev.ptr, view.nbytes), 'd2h')
            else:
                tmp = np.empty((B,) + shape, dtype=dtype)
                _lib.check(_lib.lib.vi_d2h(h, tmp.ctypes.data_as(_lib.VOIDP), dev.ptr, tmp.nbytes), 'd2h')
                dst[rr] = tmp
        step = max(1, min(len(good), 2048))
        for s in range(0, len(good), step):
            idx = np.asarray(good[s:s + step], dtype=np.int32)
            B = len(idx)
            al = {n: np.array([params[t][n] for t in idx], dtype=np.float64) for n in self.regularization_list}
            drec, dC, dH, drank = self._solve_chunk(idx, al, calccov, 'f_')
            dchi = self._buf('f_chi2', (B,))
            _lib.check(_lib.lib.vi_chi2_f64(h, B, self.P, N, self.At.ptr, dC.ptr, drec.ptr, self.dW.ptr,
                                            self.db.ptr, dchi.ptr), 'vi_chi2_f64')
            down(Coeffs, dC, s, B, (N,), np.float64)
            down(chi, dchi, s, B, (), np.float64)
            down(ranks, drank, s, B, (), np.int32)
            if calccov:
                # dC = H AWA H needs AWA of the selected records, contiguous
                dsel = self._buf('f_AWAsel', (B, N, N))
                _lib.check(_lib.lib.vi_form_system_f64(h, B, N, self.dAWA.ptr, drec.ptr, None, None, dsel.ptr),
                           'vi_form_system_f64')
                if defer_cov and not compact:
                    dall = self._buf('fd_dC', (len(good), N, N))
                    _lib.check(_lib.lib.vi_cov_f64(h, B, N, dH.ptr, dsel.ptr, dall.offset_ptr(s * N * N)), 'vi_cov_f64')
                else:
                    ddC = self._buf('f_dC', (B, N, N))
                    _lib.check(_lib.lib.vi_cov_f64(h, B, N, dH.ptr, dsel.ptr, ddC.ptr), 'vi_cov_f64')
                    down(Cov, ddC, s, B, (N, N), np.float64)
        if compact:
            return Coeffs, Cov, chi, ranks, good
        if defer_cov and calccov and len(good):
            self.join_cov()                         # (an earlier download of this engine still under way: none, normally)
            _lib.check(_lib.lib.vi_d2h_side_mark(h), 'vi_d2h_side_mark')
            dall = self._buf('fd_dC', (len(good), N, N))
            # runs of consecutive records: one copy each, straight into the rows
            cut = np.nonzero(np.diff(rows) != 1)[0] + 1
            runs = list(zip(np.concatenate([[0], cut]).tolist(), np.concatenate([cut, [len(rows)]]).tolist()))
            err = []

            def bring():
                try:
                    for a, b in runs:
                        if Cov.flags['C_CONTIGUOUS']:
                            view = Cov[int(rows[a]):int(rows[a]) + (b - a)]
                            _lib.check(_lib.lib.vi_d2h_side(h, view.ctypes.data_as(_lib.VOIDP), dall.offset_ptr(a * N * N),
                                                            view.nbytes), 'vi_d2h_side')
                        else:
                            tmp = np.empty((b - a, N, N))
                            _lib.check(_lib.lib.vi_d2h_side(h, tmp.ctypes.data_as(_lib.VOIDP), dall.offset_ptr(a * N * N),
                                                            tmp.nbytes), 'vi_d2h_side')
                            Cov[rows[a:b]] = tmp
                except BaseException as e:          # re-raised by join_cov in the caller's thread
                    err.append(e)
            th = threading.Thread(target=bring)
            th.start()
            self._cov_pending = (th, err)
            self.stats['cov_side_downloads'] = self.stats.get('cov_side_downloads', 0) + 1
        return Coeffs, Cov, chi, ranks

    def join_cov(self):
        """Wait for the covariance download a finalize(defer_cov=True) left running."""
        pend, self._cov_pending = getattr(self, '_cov_pending', None), None
        if pend is not None:
            pend[0].join()
            self.stats['cov_side_joined'] = self.stats.get('cov_side_joined', 0) + 1
            if pend[1]:
                raise pend[1][0]

    CONSISTENCY_TOL = 1e-6        # |chi^2_final - nu| <= tol * nu: the record is reported as consistent
    REDO_TOL = 1e-4               # beyond this the record's root search is redone with cold solves only

    def _search_and_finalize(self, npts, calccov, prefetch, multisection, out=None):
        """chi^2 search + final solve, with a consistency guard between the two.

        The root finder's iterates are served from each record's rotated system (warm start), the final
        coefficients from a cold solve of the untransformed system.  Where X(alpha) has eigenvalues at the
        truncation threshold the two can disagree about which of them survive (measured at the default order,
        N = 144, golden fit_default record 1: the warm search declared chi^2 = nu = 495 at an alpha where the cold
        solve gives 497.42, because chi^2(alpha) jumps by ~3 there), and the alpha the warm search returns is then
        not a root of the function the final solve evaluates.  So after the final solve every 'root' record is
        checked: |chi^2_final - nu| <= 1e-6 nu marks it consistent (search info 'consistent', 'chi2_minus_nu'); a
        record off by more than 1e-4 nu has its search redone with cold solves only and is finalised again.  If it
        still misses nu, chi^2(alpha) - nu has no root there but a sign-changing jump, which is what Brent converges
        to in the reference as well (its own golden records end at chi^2 = 540.9 and 526.2 for nu = 550, and its
        final chi^2 misses nu by 1e-4 .. 7e-4 on most default-order records); the record keeps the cold result.
        Between the two tolerances lies the noise of the warm transform itself (1e-6 .. 2e-6 of nu at N = 144, where
        the fit is only reproducible to 1e-3 anyway); redoing those cold would triple the cost of a batch for
        nothing."""
        stamp = self._stage_stamp
        stamp(None)
        params, infos = self.search(npts, prefetch=prefetch, multisection=multisection)
        stamp('search')
        # A fit that runs as ONE chain has its covariances come down beside the stream while the guard's solves run (1000
        # records: 341 -> 331 ms); the pipelines of a split batch fill each other's waits already, and a download thread each
        # only adds to the host's load there (10 000 records in four pipelines: 2331 ms in line, 2400 beside).
        # VINTERP_ASYNC_COV=0 / 1 forces either.
        e = os.environ.get('VINTERP_ASYNC_COV', 'auto')
        Coeffs, Cov, chi, ranks = self.finalize(params, calccov=calccov, out=out,
                                                defer_cov=(e == '1' or (e != '0' and out is None)))
        stamp('finalize')
        cov_rows = []                           # (record, covariance) the guard replaces: written once the download is in
        try:
            res = self._guard(npts, calccov, prefetch, params, infos, Coeffs, Cov, chi, ranks, cov_rows, stamp)
        except BaseException as guard_exc:
            # the download thread is joined either way, but the guard's own failure is the one the caller must see: a failure
            # of the thread on top of it travels as its context, not in its place
            try:
                self.join_cov()
            except BaseException as down_exc:
                raise guard_exc from down_exc
            raise
        self.join_cov()
        for t, row in cov_rows:
            Cov[t] = row
        return res

    def _guard(self, npts, calccov, prefetch, params, infos, Coeffs, Cov, chi, ranks, cov_rows, stamp):
        """The consistency guard of _search_and_finalize; Cov is not touched here (its download may still be running): rows
        to replace go to cov_rows."""
        if len(self.regularization_list) != 1 or os.environ.get('VINTERP_GUARD', '1') == '0':
            return params, infos, Coeffs, Cov, chi, ranks
        name = self.regularization_list[0]
        inf = infos[name]

        def violators():
            bad = []
            for t in range(self.T):
                if inf['outcomes'][t] != 'root':
                    continue
                nu = inf['info'][t]['sf'] * npts[t]
                inf['info'][t]['chi2_minus_nu'] = float(chi[t] - nu)
                inf['info'][t]['consistent'] = bool(abs(chi[t] - nu) <= self.CONSISTENCY_TOL * nu)
                if not abs(chi[t] - nu) <= self.REDO_TOL * nu:
                    bad.append(t)
            return bad
        bad = violators()
        inf['polished_cold'] = []
        if bad and self.warm_enabled():
            # ONE launch of cold solves for all suspect records (round 4; until then the far end, the scan and every polish
            # round were launches of their own, each as long as a cold solve - 20 of the 42 ms of the bench's jump record):
            #  * the far end of Brent's final bracket (root, other_end).  Is the miss a jump of the cold function itself?  If
            #    the COLD chi^2 - nu changes sign across that bracket - 2e-12 wide, or up to 1e-7 where the iteration ended on
            #    the jump rule - the cold function has a sign-changing jump there (an eigenvalue of X(alpha) crossing the
            #    truncation threshold or zero) and the root is as good an answer for it as for the warm one: no redo;
            #  * otherwise the warm search was misled (its chi^2 differs from the cold one by more than REDO_TOL there - next
            #    to the poles and jumps of chi^2(alpha) the rotated system is not accurate enough; the cold function's jump
            #    sits 1e-6 decades from the warm one's in the median).  The warm root is still close to a root or jump of the
            #    cold function: a sign change of the COLD chi^2 - nu within 1e-6 .. 1e-1 decades of it (12 cold solves per
            #    record) gives a small bracket for alpha_search.run_polish_batched.
            deltas = np.array([-1e-1, -1e-2, -1e-3, -1e-4, -1e-5, -1e-6, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1])
            roots = np.array([inf['info'][t]['log10_alpha'] for t in bad])
            has_oe = [inf['info'][t].get('other_end') is not None for t in bad]
            rec_l, xs_l = [], []
            for i, t in enumerate(bad):
                if has_oe[i]:
                    rec_l.append(t)
                    xs_l.append(inf['info'][t]['other_end'])
                rec_l += [t] * len(deltas)
                xs_l += (roots[i] + deltas).tolist()
            cv = self.chi2_batch(np.asarray(rec_l, dtype=np.int32), {name: np.power(10., np.asarray(xs_l, dtype=np.float64))})
            brackets, skip_brent = {}, set()
            o = 0
            for i, t in enumerate(bad):
                nu = inf['info'][t]['sf'] * npts[t]
                if has_oe[i]:
                    c = cv[o]
                    o += 1
                    if (c - nu) * (chi[t] - nu) < 0:
                        inf['info'][t]['jump'] = True
                        inf['info'][t]['chi2_other_end_minus_nu'] = float(c - nu)
                cvals = cv[o:o + len(deltas)]
                o += len(deltas)
                if inf['info'][t].get('jump'):
                    continue
                lo, hi = inf['info'][t]['bracket']
                lo, hi = min(lo, hi), max(lo, hi)
                pts = sorted([(float(roots[i] + d), float(c - nu)) for d, c in zip(deltas, cvals) if lo <= roots[i] + d <= hi]
                             + [(float(roots[i]), float(chi[t] - nu))])
                best = None
                for (xa, fa), (xb, fb) in zip(pts[:-1], pts[1:]):
                    if np.isfinite(fa) and np.isfinite(fb) and fa * fb < 0:
                        dist = max(abs(xa - roots[i]), abs(xb - roots[i]))
                        if best is None or dist < best[0]:
                            best = (dist, xa, xb, fa, fb)
                if best is not None:
                    brackets[t] = best[1:]
                    # the signature of a jump (or pole): a sign change on a short bracket whose ends both miss nu by more
                    # than REDO_TOL nu - a root there would need a slope of 2 nu per 1e-4 decades.  Brent could only bisect it.
                    if (abs(best[2] - best[1]) <= 1e-4 * (1. + 1e-9) and min(abs(best[3]), abs(best[4])) > self.REDO_TOL * nu):
                        skip_brent.add(t)
            bad = [t for t in bad if not inf['info'][t].get('jump')]
            if brackets:
                nus = {t: inf['info'][t]['sf'] * npts[t] for t in brackets}

                def f_batch(r, x):
                    return self.chi2_batch(r, {name: np.power(10., x)}) - np.array([nus[int(t)] for t in r])
                # stop at |chi^2 - nu| <= CONSISTENCY_TOL nu, or when the sign change is confined to 1e-7 decades (a jump):
                # every round is a launch as long as one cold solve, and alpha means nothing beyond ~1e-6 decades here
                root_of = {t: float(inf['info'][t]['log10_alpha']) for t in brackets}
                sol = alpha_search.run_polish_batched(brackets, f_batch, {t: self.CONSISTENCY_TOL * nus[t] for t in brackets},
                                                      target=root_of, skip_brent=skip_brent)
                done = sorted(sol)
                for t in done:
                    root, iters, oe, how = sol[t]
                    params[t][name] = float(np.power(10., root))
                    inf['info'][t].update(log10_alpha=root, other_end=oe, polished_cold=True, polish_iterations=iters,
                                          polish_end=how, warm_log10_alpha=root_of[t])
                C2, V2, c2, r2, g2 = self.finalize(params, calccov=calccov, only=set(done), compact=True)
                at = {t: i for i, t in enumerate(g2)}
                for t in done:
                    i = at.get(t)
                    Coeffs[t], chi[t], ranks[t] = (C2[i], c2[i], r2[i]) if i is not None else (np.nan, np.nan, -1)
                    if calccov:
                        cov_rows.append((t, V2[i] if i is not None else np.nan))
                inf['polished_cold'] = done
                for t in violators():
                    if t in sol:
                        # Brent on the cold function keeps a sign change inside its bracket: what is left is a jump
                        inf['info'][t]['jump'] = True
                bad = [t for t in bad if t not in sol]
        inf['redone_cold'] = list(bad)
        stamp('guard')
        if bad and self.warm_enabled():
            p2, i2 = self.search(npts, prefetch=prefetch, only=bad, cold=True)
            C2, V2, c2, r2, g2 = self.finalize(p2, calccov=calccov, only=set(bad), compact=True)
            at = {t: i for i, t in enumerate(g2)}
            for t in bad:
                params[t] = p2[t]
                inf['outcomes'][t] = i2[name]['outcomes'][t]
                inf['info'][t] = i2[name]['info'][t]
                inf['info'][t]['redone_cold'] = True
                i = at.get(t)
                Coeffs[t], chi[t], ranks[t] = (C2[i], c2[i], r2[i]) if i is not None else (np.nan, np.nan, -1)
                if calccov:
                    cov_rows.append((t, V2[i] if i is not None else np.nan))
            inf['evaluations'] += i2[name]['evaluations']
            violators()
        return params, infos, Coeffs, Cov, chi, ranks

    def _stage_stamp(self, name):
        """VINTERP_STAGE_TIMES=1: wall time per stage of a fit (device drained at the boundaries) into stats['ms_<stage>']."""
        if os.environ.get('VINTERP_STAGE_TIMES') != '1':
            return
        import time
        self.ctx.sync()
        now = time.perf_counter()
        if name is not None:
            self.stats['ms_' + name] = self.stats.get('ms_' + name, 0.) + (now - self._stage_t0) * 1e3
        self._stage_t0 = now

    def result_buffers(self, calccov=True, pinned=True):
        """Arrays for fit_resident(out=...): (Coeffs (T, N), Covariance (T, N, N) or None, chi_sq (T,), ranks (T,) int32) for the
        resident records, in page-locked host memory unless pinned=False.  A caller that fits batch after batch of the same size
        (one shard of timesteps after the other, results written out in between) hands the same arrays to every fit instead of
        receiving 8 T N^2 bytes of fresh pages each time - 1.66 GB per 10 000 records at N = 144, whose first touch and release
        cost a fit of 2.1 s about 0.1 s - and the covariances come down at the rate of the link."""
        T, N = self.T, self.N
        mk = _lib.pinned_empty if pinned else np.empty
        return (mk((T, N)), mk((T, N, N)) if calccov else None, mk((T,)), mk((T,), np.int32))

    def _check_out(self, out, calccov):
        T, N = self.T, self.N
        want = (((T, N), np.float64), ((T, N, N), np.float64), ((T,), np.float64), ((T,), np.int32))
        if len(out) != 4:
            raise ValueError('out: (Coeffs, Covariance, chi_sq, ranks)')
        for i, (a, (shape, dt)) in enumerate(zip(out, want)):
            if i == 1 and not calccov:
                continue
            if not isinstance(a, np.ndarray) or a.shape != shape or a.dtype != dt or not a.flags['C_CONTIGUOUS'] \
                    or not a.flags['WRITEABLE']:
                raise ValueError('out[%d]: a writeable C-contiguous %s array of shape %s expected' % (i, np.dtype(dt).name, shape))
        return (out[0], out[1] if calccov else None, out[2], out[3])

    def fit_resident(self, npts, calccov=True, prefetch=None, multisection=None, _out=None, out=None):
        """Fit the records made resident by upload_records().  out: the caller's result arrays (result_buffers()), written in
        place and returned in the result instead of new ones."""
        if out is not None:
            _out = self._check_out(out, calccov)
        if len(self._bounds) > 2:
            return self._fit_pipelined(npts, calccov, prefetch, multisection, _out)
        self._stage_stamp(None)
        self.form_normal_equations()
        self._stage_stamp('normal_equations')
        params, infos, Coeffs, Cov, chi, ranks = self._search_and_finalize(npts, calccov, prefetch, multisection, out=_out)
        return dict(Coeffs=Coeffs, Covariance=Cov, chi_sq=chi, reg_params=params, ranks=ranks, search=infos)

    def _fit_pipelined(self, npts, calccov, prefetch, multisection, out=None):
        """The batch as K independent sub-batches, each driven by its own host thread on its own context (stream, rocBLAS
        handle, workspace).  A fit is a chain of ~80 dependent launches with host logic in between, and a launch lasts as
        long as its slowest system: one pipeline leaves the GPU idle a quarter of the time and half empty for much of the
        rest; the launches of independent sub-batches fill those gaps (measured, 26 x 100 geometry: 1000 records 1280 ->
        1430 records/s with three pipelines, 4000 records 1330 -> 1670 with four).  Records are independent and a record's
        numbers do not depend on the batch it is in (test_c1_fit_is_independent_of_the_batch_and_consistent), so the
        split changes nothing but the time."""
        import threading
        T, N, K = self.T, self.N, len(self._bounds) - 1
        if self._subs is None or len(self._subs) != K:
            self._close_subs()
            self._subs = []
            for k in range(K):
                ctx = self.ctx if k == 0 else _lib.Context(self.ctx.device)
                sub = FitEngine(ctx, self.At, self.P, N, {n: self._R_host[n] for n in self.regularization_list},
                                self.regularization_list)
                sub._no_pipeline = True
                self._subs.append(sub)
        if out is not None:
            Coeffs, Cov, chi, ranks = out
        else:
            Coeffs = np.empty((T, N))
            Cov = np.empty((T, N, N)) if calccov else None
            chi = np.empty(T)
            ranks = np.empty(T, dtype=np.int32)
        results, errors = [None] * K, [None] * K

        def run(k):
            lo, hi = self._bounds[k], self._bounds[k + 1]
            try:
                sub = self._subs[k]
                sub.adopt_records(_View(self.dW, lo * self.P), _View(self.db, lo * self.P), hi - lo, *self._ref_host[k])
                sub.stats = dict(solves=0, launches=0)
                out = (Coeffs[lo:hi], Cov[lo:hi] if calccov else None, chi[lo:hi], ranks[lo:hi])
                results[k] = sub.fit_resident(list(npts[lo:hi]), calccov, prefetch, multisection, _out=out)
                sub.ctx.sync()
            except BaseException as e:          # re-raised in the caller's thread
                errors[k] = e
        threads = [threading.Thread(target=run, args=(k,)) for k in range(1, K)]
        for th in threads:
            th.start()
        run(0)
        for th in threads:
            th.join()
        for e in errors:
            if e is not None:
                raise e
        params, infos = [], {}
        for k, r in enumerate(results):
            lo = self._bounds[k]
            params += r['reg_params']
            for name, inf in r['search'].items():
                m = infos.setdefault(name, dict(outcomes=[], info=[], evaluations=0, polished_cold=[], redone_cold=[]))
                m['outcomes'] += inf['outcomes']
                m['info'] += inf['info']
                m['evaluations'] += inf.get('evaluations', 0)
                m['polished_cold'] += [lo + t for t in inf.get('polished_cold', [])]
                m['redone_cold'] += [lo + t for t in inf.get('redone_cold', [])]
            for key, v in self._subs[k].stats.items():
                self.stats[key] = self.stats.get(key, 0) + v
        self.stats['pipelines'] = K
        return dict(Coeffs=Coeffs, Covariance=Cov, chi_sq=chi, reg_params=params, ranks=ranks, search=infos)

    def fit(self, W, b, npts, calccov=True, prefetch=None, multisection=None, method='chi2', point_lists=None):
        self.upload_records(W, b)
        if method == 'gcv':
            self.form_normal_equations()
            params, infos = self.search_gcv(point_lists)
            Coeffs, Cov, chi, ranks = self.finalize(params, calccov=calccov)
            return dict(Coeffs=Coeffs, Covariance=Cov, chi_sq=chi, reg_params=params, ranks=ranks, search=infos)
        return self.fit_resident(npts, calccov=calccov, prefetch=prefetch, multisection=multisection)

    def solve_timing(self, enable=-1):
        """Context.solve_timing over every context this engine drives (its own and those of its pipelines): sums, except
        max_ms.  With several pipelines the launches overlap in time, so total_ms exceeds the wall time they took."""
        ctxs = [self.ctx] + [sub.ctx for sub in (self._subs or [])[1:]]
        tot = None
        for cx in ctxs:
            st = cx.solve_timing(enable)
            if tot is None:
                tot = dict(st)
            else:
                for k, v in st.items():
                    tot[k] = max(tot[k], v) if k == 'max_ms' else tot[k] + v
        return tot

    def _close_subs(self):
        for k, sub in enumerate(self._subs or []):
            sub.close()
            if k > 0:
                sub.ctx.close()
        self._subs = None

    def __del__(self):
        try:                                    # the contexts of the pipelines own streams and rocBLAS handles
            self.close()
        except Exception:
            pass

    def close(self):
        self._close_subs()
        for b in self._bufs.values():
            b.free()
        self._bufs = {}
        for r in self.R.values():
            r.free()
        self.R = {}
