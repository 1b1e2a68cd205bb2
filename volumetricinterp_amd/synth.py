"""Deterministic synthetic inputs.

The same generator feeds the golden fixtures, the CPU baseline and the GPU
runs.  The reference ships no data file (example_config.ini:9 names one that is
not in the repository), so beam geometries, records and query grids are
synthetic; ``bench.py`` says so in its ``data`` field.
"""
import numpy as np

# (beams, ranges) of the BASELINE.json configurations
GEOM_C1 = (11, 50)
GEOM_C2 = (26, 100)
GEOM_C5 = (64, 200)


def beams(nb, nr, seed=0, latcp=78.0, loncp=262.0):
    """Flat-tangent AMISR-like field of view: nb beams x nr range gates.

    Returns lat (deg), lon (deg), alt (m), each of shape (nb*nr,).
    """
    rng = np.random.default_rng(seed)
    az = np.radians(rng.uniform(0., 360., nb))
    el = np.radians(rng.uniform(35., 90., nb))
    rg = np.linspace(100e3, 700e3, nr)
    e = np.cos(el)[:, None] * np.sin(az)[:, None] * rg[None, :]
    n = np.cos(el)[:, None] * np.cos(az)[:, None] * rg[None, :]
    u = np.sin(el)[:, None] * rg[None, :]
    lat = latcp + n / 111e3
    lon = loncp + e / (111e3 * np.cos(np.radians(latcp)))
    return lat.ravel().copy(), lon.ravel().copy(), u.ravel().copy()


def synth_record(A, seed):
    """One record of model-generated 'truth' plus noise.  Returns b, err, Ctrue."""
    P, N = A.shape
    rng = np.random.default_rng(seed)
    cn = np.sqrt(np.sum(A * A, axis=0))
    Ct = rng.standard_normal(N) * 1e11 / (cn * np.sqrt(N))
    Ct[0] += 3e11 / np.mean(np.abs(A[:, 0]))
    truth = A @ Ct
    err = 0.05 * np.abs(truth) + 1e10
    b = truth + err * rng.standard_normal(P)
    return b, err, Ct


def synth_records(A, T, seed0=1000):
    """T records, timestep t seeded with seed0 + t.  Returns value (T,P), error (T,P)."""
    vals, errs = [], []
    for t in range(T):
        b, e, _ = synth_record(A, seed0 + t)
        vals.append(b)
        errs.append(e)
    return np.array(vals), np.array(errs)


def unix_times(T, t0=1480286700.0, dt=60.0):
    """(T,2) start/end Unix times, 2016-11-27T22:45:00Z onwards, 1-minute records."""
    s = t0 + dt * np.arange(T)
    return np.stack([s, s + dt], axis=1)


def query_grid(n, lat=(75., 81.), lon=(250., 274.), alt=(100e3, 700e3)):
    """n^3 geodetic query grid, 'ij' indexing, each array of shape (n,n,n)."""
    return np.meshgrid(np.linspace(lat[0], lat[1], n), np.linspace(lon[0], lon[1], n),
                       np.linspace(alt[0], alt[1], n), indexing='ij')


def synthetic_reg_matrix(AWA):
    """Throughput-run stand-in for a regularisation matrix at orders with no fixture."""
    N = AWA.shape[0]
    return np.eye(N) * np.mean(np.abs(np.diag(AWA)))
