"""ctypes binding of libvinterp.so (include/vinterp.h).

The HIP library is the product path: there is NO CPU fallback.  Importing this
module without a built ``csrc/libvinterp.so`` raises, and every compute entry
point raises ``VinterpError`` when no GPU context can be created.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VINTERP_LIB') or os.path.join(_HERE, 'csrc', 'libvinterp.so')    # VINTERP_LIB: a diagnostic build


class VinterpError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        'volumetricinterp_amd: %s is missing - build it with `make -C %s` (or __graft_entry__.build()); '
        'there is no CPU fallback for the fit/evaluate path.' % (LIB_PATH, os.path.dirname(LIB_PATH)))

# The pipelines of a batched fit (fitengine.FitEngine._fit_pipelined) drive one GPU from up to four streams besides the
# context's own; the runtime multiplexes streams onto 4 hardware queues by default and a fifth stream then waits behind an
# unrelated one (measured: 1000 records in four pipelines 760 ms with 4 queues, 583 ms with 8).  Read by the HIP runtime when
# it initialises, i.e. at the first library call below; an explicit setting in the environment wins.  It is a process-wide
# setting that child processes inherit, and it has no effect when the HIP runtime was initialised before this module was
# imported (another HIP library in the process): HW_QUEUES_REQUESTED records what this import found / asked for and
# FitEngine.pipelines() falls back to two pipelines when the eight queues were not ours to ask for (README).
HW_QUEUES_PRESET = os.environ.get('GPU_MAX_HW_QUEUES')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
HW_QUEUES_REQUESTED = os.environ['GPU_MAX_HW_QUEUES']

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
ABI_VERSION = 2                 # include/vinterp.h VI_ABI_VERSION: the signatures bound below and in fitengine.py
lib.vi_abi_version.restype = C.c_int
if lib.vi_abi_version() != ABI_VERSION:
    raise ImportError('volumetricinterp_amd: %s reports ABI version %d, this package binds version %d - rebuild it '
                      '(`make -C %s`)%s' % (LIB_PATH, lib.vi_abi_version(), ABI_VERSION, os.path.join(_HERE, 'csrc'),
                                            '; VINTERP_LIB points at it' if os.environ.get('VINTERP_LIB') else ''))

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
VOIDP = C.c_void_p

VI_MODEL_SPHHARMLAG = 1
VI_MODEL_RADBASFUN = 2


class SphGroup(C.Structure):
    _fields_ = [('v0', C.c_double), ('nvmax', C.c_int32), ('nterms', C.c_int32),
                ('pick', c_int32_p), ('c', c_double_p), ('seed_pref', c_double_p), ('seed_q', c_double_p)]


class ModelDesc(C.Structure):
    _fields_ = [('kind', C.c_int32), ('nbasis', C.c_int32), ('maxk', C.c_int32), ('maxl', C.c_int32),
                ('rot_cos', C.c_double), ('rot_sin', C.c_double), ('rot_kx', C.c_double), ('rot_ky', C.c_double),
                ('earth_radius', C.c_double), ('ngroups', C.c_int32), ('groups', C.POINTER(SphGroup)),
                ('coef_scale', c_double_p), ('coef_scale1', c_double_p), ('nu', c_double_p),
                ('centers', c_double_p), ('eps', C.c_double)]


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


I64 = C.c_int64
_sig('vi_abi_version', C.c_int)
_sig('vi_device_count', C.c_int, C.POINTER(C.c_int))
_sig('vi_ctx_create', C.c_int, C.c_int, C.POINTER(VOIDP))
_sig('vi_ctx_destroy', None, VOIDP)
_sig('vi_ctx_sync', C.c_int, VOIDP)
_sig('vi_last_error', C.c_char_p)
_sig('vi_dmalloc', C.c_int, VOIDP, C.c_size_t, C.POINTER(VOIDP))
_sig('vi_dfree', C.c_int, VOIDP, VOIDP)
_sig('vi_h2d', C.c_int, VOIDP, VOIDP, VOIDP, C.c_size_t)
_sig('vi_d2h', C.c_int, VOIDP, VOIDP, VOIDP, C.c_size_t)
_sig('vi_dmemset', C.c_int, VOIDP, VOIDP, C.c_int, C.c_size_t)
_sig('vi_d2h_side_mark', C.c_int, VOIDP)
_sig('vi_d2h_side', C.c_int, VOIDP, VOIDP, VOIDP, C.c_size_t)
_sig('vi_mem_info', C.c_int, VOIDP, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t))
_sig('vi_timer_start', C.c_int, VOIDP)
_sig('vi_timer_stop_ms', C.c_int, VOIDP, c_double_p)
_sig('vi_model_create', C.c_int, VOIDP, C.POINTER(ModelDesc), C.POINTER(VOIDP))
_sig('vi_model_destroy', None, VOIDP)
_sig('vi_basis_f64', C.c_int, VOIDP, I64, VOIDP, VOIDP, VOIDP, VOIDP, I64, I64)
_sig('vi_grad_basis_f64', C.c_int, VOIDP, I64, VOIDP, VOIDP, VOIDP, VOIDP, I64, I64, I64)
_sig('vi_eval_grad_f64', C.c_int, VOIDP, I64, VOIDP, VOIDP, VOIDP, VOIDP, VOIDP)
_sig('vi_eval_err_f64', C.c_int, VOIDP, I64, VOIDP, VOIDP, VOIDP, VOIDP, VOIDP)
_sig('vi_transform_f64', C.c_int, VOIDP, I64, VOIDP, VOIDP, VOIDP, VOIDP, VOIDP, VOIDP)
_sig('vi_eval_f64', C.c_int, VOIDP, I64, VOIDP, VOIDP, VOIDP, I64, VOIDP, VOIDP, C.c_int32, C.c_double, VOIDP)
_sig('vi_eval_basis_f64', C.c_int, VOIDP, I64, VOIDP, VOIDP, VOIDP, VOIDP, C.c_int32, C.c_double, VOIDP)
_sig('vi_eval_resident_f64', C.c_int, VOIDP, I64, I64, VOIDP, VOIDP, VOIDP)
_sig('vi_eval_f64_host', C.c_int, VOIDP, I64, c_double_p, c_double_p, c_double_p, I64, c_double_p, c_double_p,
     C.c_int32, C.c_double, c_double_p)

_sig('vi_eval_kernel_ms', C.c_int, VOIDP, c_double_p)
_sig('vi_ctx_set_eval_timing', C.c_int, VOIDP, C.c_int32)
_sig('vi_model_set_eval_precision', C.c_int, VOIDP, C.c_int32)
_sig('vi_host_alloc', C.c_int, C.c_size_t, C.POINTER(VOIDP))
_sig('vi_host_free', C.c_int, VOIDP)
_sig('vi_solve_timing', C.c_int, VOIDP, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
     c_double_p, c_double_p)
_sig('vi_solve_rounds', C.c_int, VOIDP, C.POINTER(C.c_int64))
_sig('vi_rccl_unique_id', C.c_int, C.c_char_p)
_sig('vi_rccl_init', C.c_int, VOIDP, C.c_int, C.c_int, C.c_char_p)
_sig('vi_rccl_bcast_f64', C.c_int, VOIDP, VOIDP, I64, C.c_int)
_sig('vi_rccl_destroy', C.c_int, VOIDP)

EXPORTS = ['vi_eval_basis_f64', 'vi_eval_resident_f64', 'vi_host_alloc', 'vi_host_free', 'vi_model_set_eval_precision', 'vi_solve_rounds', 'vi_grad_basis_f64', 'vi_eval_grad_f64', 'vi_eval_err_f64', 'vi_eval_kernel_ms', 'vi_ctx_set_eval_timing', 'vi_solve_timing', 'vi_rccl_unique_id', 'vi_rccl_init', 'vi_rccl_bcast_f64', 'vi_rccl_destroy', 'vi_abi_version', 'vi_device_count', 'vi_ctx_create', 'vi_ctx_destroy', 'vi_ctx_sync', 'vi_last_error',
           'vi_dmalloc', 'vi_dfree', 'vi_h2d', 'vi_d2h', 'vi_d2h_side_mark', 'vi_d2h_side', 'vi_dmemset', 'vi_mem_info', 'vi_timer_start', 'vi_timer_stop_ms',
           'vi_model_create', 'vi_model_destroy', 'vi_basis_f64', 'vi_transform_f64', 'vi_eval_f64',
           'vi_eval_f64_host']


def check(rc, what=''):
    if rc != 0:
        msg = lib.vi_last_error()
        raise VinterpError('%s failed (status %d): %s' % (what or 'libvinterp call', rc,
                                                            msg.decode('utf-8', 'replace') if msg else ''))


def device_count():
    n = C.c_int(0)
    check(lib.vi_device_count(C.byref(n)), 'vi_device_count')
    return n.value


class Context:
    """One per GPU (vi_ctx): owns the HIP stream, the rocBLAS handle and the fit workspace."""

    def __init__(self, device=0):
        h = VOIDP()
        check(lib.vi_ctx_create(int(device), C.byref(h)), 'vi_ctx_create(device=%d)' % device)
        self.handle = h
        self.device = int(device)

    def sync(self):
        check(lib.vi_ctx_sync(self.handle), 'vi_ctx_sync')

    def eval_timing(self, on):
        """HIP events around the evaluation kernels of every call (read with eval_kernel_ms); off by default."""
        check(lib.vi_ctx_set_eval_timing(self.handle, 1 if on else 0), 'vi_ctx_set_eval_timing')

    def eval_kernel_ms(self):
        ms = C.c_double(0.)
        check(lib.vi_eval_kernel_ms(self.handle, C.byref(ms)), 'vi_eval_kernel_ms')
        return ms.value

    def mem_info(self):
        """(free, total) bytes of the device."""
        fr, tot = C.c_size_t(0), C.c_size_t(0)
        check(lib.vi_mem_info(self.handle, C.byref(fr), C.byref(tot)), 'vi_mem_info')
        return fr.value, tot.value

    def timer_start(self):
        check(lib.vi_timer_start(self.handle), 'vi_timer_start')

    def timer_stop_ms(self):
        ms = C.c_double(0.)
        check(lib.vi_timer_stop_ms(self.handle, C.byref(ms)), 'vi_timer_stop_ms')
        return ms.value

    def solve_timing(self, enable=-1):
        """Eigen-solve kernel timing (vi_solve_timing): returns a dict for the period since the last reset."""
        n, sy, tm, rd = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        tot, mx = C.c_double(), C.c_double()
        check(lib.vi_solve_rounds(self.handle, C.byref(rd)), 'vi_solve_rounds')      # before the reset below
        check(lib.vi_solve_timing(self.handle, int(enable), C.byref(n), C.byref(sy), C.byref(tm), C.byref(tot),
                                  C.byref(mx)), 'vi_solve_timing')
        return dict(launches=n.value, systems=sy.value, timed=tm.value, total_ms=tot.value, max_ms=mx.value,
                    rounds=rd.value)

    def empty(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype=np.float64):
        a = DeviceArray(self, shape, dtype)
        check(lib.vi_dmemset(self.handle, a.ptr, 0, a.nbytes), 'vi_dmemset')
        return a

    def to_device(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype)
        a = DeviceArray(self, host.shape, host.dtype)
        a.upload(host)
        return a

    def close(self):
        if self.handle:
            lib.vi_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:                      # a context owns a stream, a rocBLAS handle, events and a grow-only device workspace
            self.close()
        except Exception:
            pass


class DeviceArray:
    """Typed view of a vi_dmalloc allocation."""

    def __init__(self, ctx, shape, dtype=np.float64):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        p = VOIDP()
        check(lib.vi_dmalloc(ctx.handle, self.nbytes, C.byref(p)), 'vi_dmalloc(%d bytes)' % self.nbytes)
        self.ptr = p

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size > self.size:
            raise ValueError('upload larger than the allocation: %d vs %d' % (host.size, self.size))
        check(lib.vi_h2d(self.ctx.handle, self.ptr, host.ctypes.data_as(VOIDP), host.nbytes), 'vi_h2d')
        return self

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(lib.vi_d2h(self.ctx.handle, out.ctypes.data_as(VOIDP), self.ptr, self.nbytes), 'vi_d2h')
        return out

    def offset_ptr(self, nelem):
        return VOIDP(self.ptr.value + int(nelem) * self.dtype.itemsize)

    def free(self):
        if self.ptr is not None and self.ctx.handle:
            lib.vi_dfree(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_ctx_lock = threading.Lock()
_default_ctx = {}


def default_device():
    for key in ('VINTERP_DEVICE', 'LOCAL_RANK'):
        v = os.environ.get(key)
        if v is not None and v != '':
            return int(v)
    return 0


def get_context(device=None):
    """Process-wide context of a device (created on first use)."""
    if device is None:
        device = default_device()
    with _ctx_lock:
        ctx = _default_ctx.get(device)
        if ctx is None:
            n = device_count()
            if n == 0:
                raise VinterpError('no HIP device visible: the volumetricinterp_amd fit/evaluate path runs on '
                                   'an MI355X only (there is no CPU fallback)')
            ctx = Context(device % n)
            _default_ctx[device] = ctx
        return ctx


class _PinnedOwner(object):
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr is not None:
                lib.vi_host_free(self.ptr)
        except Exception:
            pass
        self.ptr = None


def pinned_empty(shape, dtype=np.float64):
    """An uninitialised ndarray in page-locked host memory (vi_host_alloc): grids and outputs kept in such arrays move
    to and from the GPU at the full rate of the link, both directions at once, inside Estimate.__call__ /
    evaluate_coeffs(out=...).  The memory is released when the last view of the array is gone."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64))
    p = VOIDP()
    check(lib.vi_host_alloc(n * dtype.itemsize, C.byref(p)), 'vi_host_alloc')
    buf = (C.c_byte * max(1, n * dtype.itemsize)).from_address(p.value)
    buf._vi_owner = _PinnedOwner(p)               # every view keeps `buf` (its .base chain) and with it the owner alive
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
