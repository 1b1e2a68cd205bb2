#!/usr/bin/env python3
"""BASELINE configs[4] "fp32 vs fp64 tolerance sweep": what an fp32 variant of the evaluation would cost in accuracy.

CPU emulation of the device recurrence (tests/emulate.py, driven by the same host tables) with the Legendre degree
recurrence - the 70 % of the kernel an fp32 variant would move to single precision - carried in float32 while
coordinates, seeds' inputs, Laguerre factors and the contraction stay in float64.  Reports, per order, the relative
error of the basis columns and of an evaluated density against the all-fp64 run.  No GPU needed."""
import io
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import emulate

CFG = '[DEFAULT]\n[MODEL]\nNAME = sphharmlag\nMAXK = %d\nMAXL = %d\nCAP_LIM = %g\nMAX_Z_INT = INF\nLATCP = 78\nLONCP = 262\n'


def tables(maxk, maxl, cap):
    # the host table builder needs no GPU (the device handle is only created on first use)
    from volumetricinterp_amd.models.sphharmlag import Model
    m = Model(io.StringIO(CFG % (maxk, maxl, cap)))
    return m.device_tables()


def run(maxk, maxl, cap, chain_dtype, lat, lon, alt):
    tb = tables(maxk, maxl, cap)
    if chain_dtype is np.float32:
        # float32 chain: cast the geometry that feeds the recurrence and the table; emulate.py keeps whatever dtype
        # its arrays have, so wrap its geometry
        orig = emulate.geom

        def geom32(tb_, la, lo, al):
            g = orig(tb_, la, lo, al)
            g['x'] = g['x'].astype(np.float32)
            g['s'] = g['s'].astype(np.float32)
            return g
        emulate.geom = geom32
        for G in tb['groups']:
            G['c'] = G['c'].astype(np.float32)
        try:
            A = emulate.basis_from_tables(tb, maxk, maxl, lat, lon, alt)
        finally:
            emulate.geom = orig
        return A
    return emulate.basis_from_tables(tb, maxk, maxl, lat, lon, alt)


def main():
    rng = np.random.default_rng(0)
    P = 4000
    lat, lon, alt = rng.uniform(75, 81, P), rng.uniform(250, 274, P), rng.uniform(100e3, 700e3, P)
    print('order (MAXK, MAXL, CAP_LIM)   N    max column rel. L2 err   density rel. L2 err (random C ~ 1/|column|)')
    for maxk, maxl, cap in [(4, 6, 10.), (8, 2, 10.), (4, 6, 15.), (8, 12, 15.)]:
        A64 = run(maxk, maxl, cap, np.float64, lat, lon, alt)
        A32 = run(maxk, maxl, cap, np.float32, lat, lon, alt)
        ok = np.all(np.isfinite(A64), axis=0) & (np.linalg.norm(A64, axis=0) > 0)
        cn = np.linalg.norm(A64[:, ok], axis=0)
        colerr = np.linalg.norm((A32 - A64)[:, ok], axis=0) / cn
        C = rng.standard_normal(ok.sum()) / cn
        d64, d32 = A64[:, ok] @ C, A32[:, ok] @ C
        print('(%d, %2d, %4.1f)                %5d   %.2e                 %.2e' %
              (maxk, maxl, cap, A64.shape[1], colerr.max(), np.linalg.norm(d32 - d64) / np.linalg.norm(d64)))


if __name__ == '__main__':
    main()
