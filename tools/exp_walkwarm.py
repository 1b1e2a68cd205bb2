"""Diagnostic: is a warm start from the alpha -> 0 eigenbasis worth it for the bracket walk (alpha = 1e-20 .. 1e-101)?"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden, rel
from volumetricinterp_amd import _lib, fitengine
ctx = _lib.get_context(); h = ctx.handle
f = load_golden('fit_default'); R = f['R']; AWA = f['rec0_AWA']; y = f['rec0_y']; N = 144
EPS = np.finfo(float).eps
nrec = 64                                     # 64 copies of the record -> fill the chip like a batch would
dAWA = ctx.to_device(np.stack([AWA] * nrec)); dy = ctx.to_device(np.stack([y] * nrec)); dR = ctx.to_device(R)
a0 = float(sys.argv[1]) if len(sys.argv) > 1 else -60.
rec = np.arange(nrec, dtype=np.int32)
drec = ctx.to_device(rec); dal0 = ctx.to_device(np.full(nrec, 10.0**a0))
dC0 = ctx.empty((nrec, N)); drk = ctx.empty((nrec,), np.int32)
dV = ctx.empty((nrec, N, N)); dD1 = ctx.empty((nrec, N, N)); dD2 = ctx.empty((nrec, N, N)); dyt = ctx.empty((nrec, N))
ctx.timer_start()
_lib.check(_lib.lib.vi_warm_prepare_f64(h, nrec, N, dAWA.ptr, drec.ptr, dal0.ptr, dR.ptr, dy.ptr, EPS, dC0.ptr, drk.ptr,
                                        dV.ptr, dD1.ptr, dD2.ptr, dyt.ptr), 'prep')
print('prepare %d records at 1e%g: %.2f ms' % (nrec, a0, ctx.timer_stop_ms()))
alist = np.arange(-21, -102, -1.0)
for name, arr in (('all walk alphas', alist), ('-21..-30', alist[:10]), ('-31..-60', alist[10:40]), ('-61..-101', alist[40:])):
    B = nrec * len(arr)
    recs = np.repeat(rec, len(arr)).astype(np.int32); al = np.tile(10.0**arr, nrec)
    dr = ctx.to_device(recs); da = ctx.to_device(al)
    dCw = ctx.empty((B, N)); dCc = ctx.empty((B, N)); drk2 = ctx.empty((B,), np.int32); dX = ctx.empty((B, N, N))
    ctx.timer_start()
    _lib.check(_lib.lib.vi_warm_solve_f64(h, B, N, dD1.ptr, dD2.ptr, dyt.ptr, dV.ptr, dr.ptr, da.ptr, EPS, dCw.ptr, drk2.ptr), 'warm')
    tw = ctx.timer_stop_ms()
    ctx.timer_start()
    _lib.check(_lib.lib.vi_form_system_f64(h, B, N, dAWA.ptr, dr.ptr, da.ptr, dR.ptr, dX.ptr), 'form')
    _lib.check(_lib.lib.vi_solve_trunc_f64(h, B, N, dX.ptr, dy.ptr, dr.ptr, EPS, dCc.ptr, drk2.ptr, N * EPS, None), 'cold')
    tc = ctx.timer_stop_ms()
    print('%-16s B=%5d: warm %.2f ms (%.1f us/system)   cold %.2f ms (%.1f us/system)' % (name, B, tw, tw * 1e3 / B, tc, tc * 1e3 / B))
