"""WGS84 geodetic -> ECEF on the host (pymap3d.geodetic2ecef; the reference calls it at
models/sphharmlag.py:345, interpolate.py:422, estimate.py:172).  Host one-offs only (rotation
constants, RBF centres, convex hull); per-point work uses the device version in csrc/vi_basis.hip."""
import numpy as np

WGS84_A = 6378137.0
WGS84_B = 6356752.31424518


def geodetic2ecef(lat, lon, alt):
    lat = np.radians(np.asarray(lat, dtype=np.float64))
    lon = np.radians(np.asarray(lon, dtype=np.float64))
    alt = np.asarray(alt, dtype=np.float64)
    N = WGS84_A**2 / np.sqrt(WGS84_A**2 * np.cos(lat)**2 + WGS84_B**2 * np.sin(lat)**2)
    x = (N + alt) * np.cos(lat) * np.cos(lon)
    y = (N + alt) * np.cos(lat) * np.sin(lon)
    z = (N * (WGS84_B / WGS84_A)**2 + alt) * np.sin(lat)
    return x, y, z
