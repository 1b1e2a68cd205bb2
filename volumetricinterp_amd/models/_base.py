"""What the two model plug-ins share on the device side: the vi_model handle and the basis / transform calls."""
import numpy as np

from .. import _lib


class DeviceModel(object):
    """Mixin for models/<NAME>.Model.  Subclasses implement ``_create_handle(ctx) -> (handle, keepalive)``."""

    _ctx = None
    _handle = None
    _keep = None
    nbasis = 0

    def handle(self, ctx=None):
        """Create (once) the device-resident model and return its vi_model handle."""
        if self._handle is not None:
            return self._handle
        if ctx is not None:
            self._ctx = ctx
        if self._ctx is None:
            self._ctx = _lib.get_context()
        self._handle, self._keep = self._create_handle(self._ctx)
        return self._handle

    @property
    def ctx(self):
        self.handle()
        return self._ctx

    def set_eval_precision(self, chains='f64'):
        """Arithmetic of the Legendre degree recurrences of the fused evaluation (vi_model_set_eval_precision):
        'f64' (default, the reference's float64) or 'f32' - fp32 chains, everything else fp64 (BASELINE configs[4]'s
        tolerance sweep; it misses the 1e-6 tolerance)."""
        if chains not in ('f64', 'f32'):
            raise ValueError("chains must be 'f64' or 'f32'")
        _lib.check(_lib.lib.vi_model_set_eval_precision(self.handle(), 1 if chains == 'f32' else 0),
                   'vi_model_set_eval_precision')

    def __del__(self):
        try:
            if self._handle is not None and self._ctx is not None and self._ctx.handle:
                _lib.lib.vi_model_destroy(self._handle)
        except Exception:
            pass
        self._handle = None

    def _upload_coords(self, gdlat, gdlon, gdalt):
        ctx = self.ctx
        return tuple(ctx.to_device(np.asarray(a, dtype=np.float64).ravel()) for a in (gdlat, gdlon, gdalt))

    def _transform(self, gdlat, gdlon, gdalt):
        """Three (P,) arrays: (z, theta, phi) for sphharmlag, ECEF (x, y, z) for radbasfun."""
        h = self.handle()
        P = np.asarray(gdlat).size
        dlat, dlon, dalt = self._upload_coords(gdlat, gdlon, gdalt)
        out = [self._ctx.empty(P) for _ in range(3)]
        _lib.check(_lib.lib.vi_transform_f64(h, P, dlat.ptr, dlon.ptr, dalt.ptr, out[0].ptr, out[1].ptr, out[2].ptr),
                   'vi_transform_f64')
        return [o.download() for o in out]

    def basis_device(self, dlat, dlon, dalt, P, transposed=False):
        """A on the device: (P, N) row-major, or the N x P layout the fit kernels consume."""
        h = self.handle()
        N = self.nbasis
        dA = self._ctx.empty((N, P) if transposed else (P, N))
        ld_p, ld_n = (1, P) if transposed else (N, 1)
        _lib.check(_lib.lib.vi_basis_f64(h, P, dlat.ptr, dlon.ptr, dalt.ptr, dA.ptr, ld_p, ld_n), 'vi_basis_f64')
        return dA

    def basis(self, gdlat, gdlon, gdalt):
        """Model.basis of the reference (sphharmlag.py:118-145, radbasfun.py:83-112): shape + (nbasis,)."""
        gdlat = np.asarray(gdlat, dtype=np.float64)
        P = gdlat.size
        if P == 0:
            return np.zeros(gdlat.shape + (self.nbasis,))
        dlat, dlon, dalt = self._upload_coords(gdlat, gdlon, gdalt)
        A = self.basis_device(dlat, dlon, dalt, P).download()
        return A.reshape(gdlat.shape + (self.nbasis,))
