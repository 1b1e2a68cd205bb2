#!/usr/bin/env python3
"""LDS bank-conflict model of one K3 round (csrc/vi_jacobi_device.h: jacobi_system) - host-side arithmetic only.

Rebuilds the kernel's LDS addresses for every lane of every wave (N = 144: 640 threads, 630 super-blocks) and prices each
LDS instruction of a cross-pair round by the bank rules of MI355X_MICROARCH.md (LDS section):
  ds_read_b64   2 groups of 32 lanes,           bank = (byte/4) mod 64
  ds_read_b128  4 groups of 16 lanes (listed),  bank = (byte/4) mod 64
  ds_write_b64  4 groups of 16 contiguous lanes, bank = (byte/4) mod 32
  ds_write_b128 8 groups of 8 contiguous lanes,  bank = (byte/4) mod 32
A group costs max over banks of the number of DISTINCT addresses on that bank (>= 1); the sum over groups minus the number of
groups is what SQ_LDS_BANK_CONFLICT counts, the sum itself what SQ_LDS_IDX_ACTIVE counts.
Usage: k3_bank_sim.py [N] [layout]      layout: 'r3' (round 3) or 'r4'
"""
import sys
from collections import defaultdict

import numpy as np

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def cost(byte_addrs, width, kind):
    """LDS cycles of one wave instruction; byte_addrs: per lane byte address or None (inactive lane)."""
    if kind == 'read' and width == 8:
        groups, nb = [list(range(0, 32)), list(range(32, 64))], 64
    elif kind == 'read' and width == 16:
        groups, nb = B128_GROUPS, 64
    elif kind == 'write' and width == 8:
        groups, nb = [list(range(16 * g, 16 * g + 16)) for g in range(4)], 32
    elif kind == 'write' and width == 16:
        groups, nb = [list(range(8 * g, 8 * g + 8)) for g in range(8)], 32
    else:
        raise ValueError
    tot = base = 0
    for grp in groups:
        banks = defaultdict(set)
        act = False
        for l in grp:
            a = byte_addrs[l]
            if a is None:
                continue
            act = True
            for d in range(width // 4):
                banks[((a // 4) + d) % nb].add((a // 4 + d) // nb)      # distinct bank rows on one bank
        if act:
            tot += max(len(v) for v in banks.values())
            base += 1
    return tot, base


def j10(r, c):
    if r == c:
        return r
    if r > c:
        r, c = c, r
    return 3 + c if r == 0 else (5 + c if r == 1 else 9)


def ring_next(s, m):
    if s == 0:
        return 0
    if s == 1:
        return 2
    if s & 1:
        return s - 2
    return 2 * m - 1 if s == 2 * m - 2 else s + 2


class Layout:
    """Element address (in doubles) of the slot-indexed matrix; r3 = the layout of round 3 (tri4)."""

    def __init__(self, N, kind):
        self.Np = (N + 3) & ~3
        self.M = self.Np // 4
        self.nsb = self.M * (self.M - 1) // 2
        self.kind = kind
        M, nsb = self.M, self.nsb
        if kind == 'r3':
            self.pstride = nsb
            self.dg = 16 * nsb
            self.dstride = M
        else:                       # r4: plane stride padded to an odd multiple of ... see choose()
            self.pstride = PSTRIDE or nsb
            self.dg = 16 * self.pstride
            self.dstride = DSTRIDE or M
        self.ntri = self.dg + 10 * self.dstride

    def kidx(self, a, b):
        return b * (b - 1) // 2 + a

    def tri4(self, i, j):
        a, r, b, c = i >> 2, i & 3, j >> 2, j & 3
        if a == b:
            return self.dg + j10(r, c) * self.dstride + a
        if a < b:
            return (4 * r + c) * self.pstride + self.kidx(a, b)
        return (4 * c + r) * self.pstride + self.kidx(b, a)

    def slot_next(self, s):
        return 2 * ring_next(s >> 1, self.M) + (s & 1)


PSTRIDE = DSTRIDE = None


def simulate(N, kind, verbose=True):
    L = Layout(N, kind)
    M, nsb, Np = L.M, L.nsb, L.Np
    NT = ((nsb + 63) // 64) * 64
    assert nsb <= 768, 'IT > 1 not modelled'
    nw = NT // 64
    yv = L.ntri
    cs = yv + 2 * Np                    # double2 array: byte address = 8 * cs + 16 * index
    tot = defaultdict(lambda: [0, 0])

    def add(name, addrs, width, k):
        t, b = cost(addrs, width, k)
        tot[name][0] += t
        tot[name][1] += b

    for w in range(nw):
        ka, kb, live = [], [], []
        for l in range(64):
            k = w * 64 + l
            lv = k < nsb
            kk = k if lv else 0
            b = int((1 + np.sqrt(1 + 8 * kk)) / 2)
            while b * (b - 1) // 2 > kk:
                b -= 1
            while (b + 1) * b // 2 <= kk:
                b += 1
            ka.append(kk - b * (b - 1) // 2 if lv else 0)
            kb.append(b if lv else 1)
            live.append(lv)
        # phase 2a: block reads (b64), 16 per lane
        for e in range(16):
            add('block read b64', [8 * (w * 64 + l + e * L.pstride) if live[l] else None for l in range(64)], 8, 'read')
        # rotation reads (b128): 4 of match a, 4 of match b
        for q in range(4):
            add('cs read b128 (a)', [8 * cs + 16 * (q * M + ka[l]) if live[l] else None for l in range(64)], 16, 'read')
            add('cs read b128 (b)', [8 * cs + 16 * (q * M + kb[l]) if live[l] else None for l in range(64)], 16, 'read')
        # phase b: block stores at the permuted slots (b64), 16 per lane
        for r in range(4):
            for c in range(4):
                addrs = []
                for l in range(64):
                    if not live[l]:
                        addrs.append(None)
                        continue
                    addrs.append(8 * L.tri4(L.slot_next(4 * ka[l] + r), L.slot_next(4 * kb[l] + c)))
                add('block store b64', addrs, 8, 'write')
        if w == 0:
            act = [l < M for l in range(64)]
            for p in range(4):
                for q in range(p, 4):
                    add('diag read b64', [8 * (L.dg + j10(p, q) * L.dstride + l) if act[l] else None for l in range(64)], 8, 'read')
                    add('diag store b64', [8 * L.tri4(L.slot_next(4 * l + p), L.slot_next(4 * l + q)) if act[l] else None
                                           for l in range(64)], 8, 'write')
            for p in range(4):
                add('y read b64', [8 * (yv + 4 * l + p) if act[l] else None for l in range(64)], 8, 'read')
                add('y store b64', [8 * (yv + Np + L.slot_next(4 * l + p)) if act[l] else None for l in range(64)], 8, 'write')
            for q in range(4):
                add('cs store b128', [8 * cs + 16 * (q * M + l) if act[l] else None for l in range(64)], 16, 'write')
    T = sum(v[0] for v in tot.values())
    Bc = sum(v[1] for v in tot.values())
    if verbose:
        print('N = %d, layout %s: plane stride %d, diagonal stride %d, %d doubles of matrix (%.1f KB)'
              % (N, kind, L.pstride, L.dstride, L.ntri, L.ntri * 8 / 1024.))
        for k, v in tot.items():
            print('  %-20s cycles %6d  conflict-free %6d  extra %6d (%.1f %%)' % (k, v[0], v[1], v[0] - v[1], 100. * (v[0] - v[1]) / max(1, v[0])))
        print('  TOTAL per round: LDS-array cycles %d, of which conflicts %d = %.1f %%' % (T, T - Bc, 100. * (T - Bc) / T))
    return T, T - Bc, L.ntri


if __name__ == '__main__':
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 144
    kind = sys.argv[2] if len(sys.argv) > 2 else 'r3'
    if len(sys.argv) > 3:
        PSTRIDE = int(sys.argv[3])
    if len(sys.argv) > 4:
        DSTRIDE = int(sys.argv[4])
    simulate(N, kind)
