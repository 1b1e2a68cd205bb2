"""Test-only NumPy emulation of the device recurrence in csrc/vi_basis.hip (sph_point), driven by the
same host tables (Model.device_tables()).  Lets the CPU suite validate the tables without a GPU.
Not part of the product path."""
import numpy as np

from oracle.geodesy import geodetic2ecef

RE = 6371.2e3


def geom(tb, lat, lon, alt):
    X, Y, Z = geodetic2ecef(lat, lon, alt)
    rc, rs, kx, ky = tb['rot_cos'], tb['rot_sin'], tb['kx'], tb['ky']
    kd = kx * X + ky * Y
    Rx = X * rc + (ky * Z) * rs + kx * kd * (1 - rc)
    Ry = Y * rc + (-kx * Z) * rs + ky * kd * (1 - rc)
    Rz = Z * rc + (kx * Y - ky * X) * rs
    rho = np.sqrt(Rx**2 + Ry**2)
    r = np.sqrt(Rx**2 + Ry**2 + Rz**2)
    x = Rz / r
    return dict(x=x, s=np.sqrt(1 - x * x), cphi=Rx / rho, sphi=Ry / rho, z=100 * (r / RE - 1))


def basis_from_tables(tb, maxk, maxl, lat, lon, alt):
    g = geom(tb, np.asarray(lat, float), np.asarray(lon, float), np.asarray(alt, float))
    P = g['x'].size
    x, s, zz = g['x'], g['s'], 0.5 * (1 - g['x'])
    cm = [np.ones(P)]
    sm = [np.zeros(P)]
    for m in range(1, maxl):
        cm.append(cm[m - 1] * g['cphi'] - sm[m - 1] * g['sphi'])
        sm.append(sm[m - 1] * g['cphi'] + cm[m - 1] * g['sphi'])
    Lk = [np.ones(P), 1 - g['z']]
    for k in range(1, maxk):
        Lk.append(((2 * k + 1 - g['z']) * Lk[k] - k * Lk[k - 1]) / (k + 1))
    E = np.exp(-0.5 * g['z'])
    L2 = maxl * maxl
    A = np.zeros((P, maxk * L2))
    for G in tb['groups']:
        cur = [np.zeros(P) for _ in range(maxl)]
        prev = [np.zeros(P) for _ in range(maxl)]
        pmm = np.ones(P)
        spow = np.ones(P)

        def hyp(q):
            r = np.ones(P)
            ssum = np.ones(P)
            for qi in q:
                r = r * qi * zz
                ssum = ssum + r
                if np.all(np.abs(r) <= 1e-17 * np.abs(ssum)):
                    break
            return ssum
        for j in range(G['nvmax'] + 1):
            for m in range(maxl):
                if j > m + 1:
                    nw = x * cur[m] - G['c'][j, m] * prev[m]
                    prev[m], cur[m] = cur[m], nw
                elif j == m:
                    if m > 0:
                        pmm = pmm * (-(2. * m - 1.)) * s
                        spow = spow * s
                    cur[m] = pmm.copy() if G['nterms'] == 0 else G['pref'][0, m] * spow * hyp(G['q'][0, m])
                elif j == m + 1:
                    prev[m] = cur[m]
                    cur[m] = (x * (2. * m + 1.) * cur[m] if G['nterms'] == 0
                              else G['pref'][1, m] * spow * hyp(G['q'][1, m]))
            l = G['pick'][j]
            if l >= 0:
                r0 = l * (l + 1)
                for m in range(l + 1):
                    for k in range(maxk):
                        with np.errstate(all='ignore'):
                            A[:, k * L2 + r0 + m] = E * Lk[k] * (tb['scale'][r0 + m] * cm[m] * cur[m])
                            if m > 0:
                                A[:, k * L2 + r0 - m] = E * Lk[k] * (tb['scale'][r0 - m] * sm[m] * cur[m])
    return A
