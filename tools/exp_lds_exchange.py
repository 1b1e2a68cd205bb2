#!/usr/bin/env python3
"""What removing K3's LDS bank conflicts could buy, measured: the block exchange of one Jacobi round
(640 threads x 16 ds_read_b64 + 16 ds_write_b64, two barriers) with the kernel's real permuted destinations against a
conflict-free (identity) store pattern - tools/microbench/lds_exchange.hip; the destinations come from the bank model
tools/sim/k3_bank_sim.py, which reproduces the profiler's conflict share (30.3 % modelled, 29 % measured).
python tools/exp_lds_exchange.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools', 'sim'))
import k3_bank_sim as sim                                          # noqa: E402

mb = C.CDLL(os.path.join(ROOT, 'tools', 'microbench', 'libldsexchange.so'))
mb.mb_lds_exchange.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]

L = sim.Layout(144, 'r3')
nsb, M = L.nsb, L.M
real = np.zeros((640, 16), dtype=np.int32)
ident = np.zeros((640, 16), dtype=np.int32)
for k in range(640):
    kk = k if k < nsb else 0
    b = int((1 + np.sqrt(1 + 8 * kk)) / 2)
    while b * (b - 1) // 2 > kk:
        b -= 1
    while (b + 1) * b // 2 <= kk:
        b += 1
    a = kk - b * (b - 1) // 2
    for r in range(4):
        for c in range(4):
            t = L.tri4(L.slot_next(4 * a + r), L.slot_next(4 * b + c))
            real[k, 4 * r + c] = t if t < 16 * nsb else kk + (4 * r + c) * nsb      # (diagonal-plane targets: keep inside the image)
            ident[k, 4 * r + c] = kk + (4 * r + c) * nsb
out = {}
for name, d in (('real permuted destinations', real), ('identity (conflict-free)', ident)):
    cyc = C.c_double()
    assert mb.mb_lds_exchange(d.ctypes.data_as(C.POINTER(C.c_int)), nsb, 2000, 256, C.byref(cyc)) == 0
    out[name] = cyc.value
    print('%-28s %.0f cycles per round (16 reads + barrier + 16 stores + barrier, 10 waves)' % (name, cyc.value))
a, b = out['real permuted destinations'], out['identity (conflict-free)']
print('the conflicts cost %.0f cycles per round = %.1f %% of the 4300 cycles of a K3 round (150 us per sweep of 72 rounds at 2.07 GHz): '
      'a perfect re-layout would take a sweep from 150 to %.0f us' % (a - b, 100. * (a - b) / 4300., 150. * (1. - (a - b) / 4300.)))
