"""Oracle (test infrastructure): WGS84 geodetic -> ECEF.

Restates ``pymap3d.geodetic2ecef`` (pymap3d >= 1.8.0, requirements.txt:1-5 of
the reference; absent from this image), which the reference calls at
models/sphharmlag.py:345,351, models/radbasfun.py:57,253, interpolate.py:422
and estimate.py:172.  Published closed form; degrees in, metres out.
"""
import numpy as np

WGS84_A = 6378137.0
WGS84_B = 6356752.31424518


def geodetic2ecef(lat, lon, alt):
    lat = np.radians(np.asarray(lat, dtype=np.float64))
    lon = np.radians(np.asarray(lon, dtype=np.float64))
    alt = np.asarray(alt, dtype=np.float64)
    a, b = WGS84_A, WGS84_B
    # prime-vertical radius of curvature
    N = a**2 / np.sqrt(a**2 * np.cos(lat)**2 + b**2 * np.sin(lat)**2)
    x = (N + alt) * np.cos(lat) * np.cos(lon)
    y = (N + alt) * np.cos(lat) * np.sin(lon)
    z = (N * (b / a)**2 + alt) * np.sin(lat)
    return x, y, z
