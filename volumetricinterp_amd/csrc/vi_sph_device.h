// Device helpers shared by the basis / evaluation kernels (vi_basis.hip, vi_eval_mfma.hip): geodetic -> model
// coordinates (sphharmlag.py:324-359), the 2F1 seed series and the Laguerre recurrence.
#pragma once
#include "vi_common.h"

namespace {

constexpr int BLOCK = 256;
constexpr double WGS84_A = 6378137.0;
constexpr double WGS84_B = 6356752.31424518;
constexpr double DEG2RAD = 0.017453292519943295;   // pi/180, the constant np.radians multiplies by

struct Geom {
    double X, Y, Z;          // ECEF, metres
    double x, s;             // cos(theta), sin(theta) of the rotated colatitude
    double cphi, sphi;       // cos/sin of the rotated azimuth
    double z;                // 100 (r/RE - 1)
    double Rx, Ry;           // rotated equatorial components (for atan2 in vi_transform)
};

// sin and cos of an angle below 1e4 rad: Cody-Waite reduction by pi/2 in two pieces (exact for |x| < 1e4: k has 13 bits) and the
// two kernels of fdlibm on |r| <= pi/4; 2.2e-16 absolute against libm on 2e7 arguments, a third of the instructions of the
// library's sincos (which carries the Payne-Hanek path for huge arguments).  Used by the hull pass's prefilter.
__device__ __forceinline__ void sincos_cw(double x, double& sn, double& cs)
{
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632679489655800e+00, x);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                               2.75573137070700676789e-06), -1.98412698298579493134e-04),
                                 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double sr = fma(r * z, ps, r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                               -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                 -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    const double a = (q & 1) ? cr : sr, b = (q & 1) ? sr : cr;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}

// pymap3d.geodetic2ecef (WGS84 closed form), called at sphharmlag.py:351 / radbasfun.py:253
__device__ __forceinline__ void geodetic2ecef(double lat, double lon, double alt, double& X, double& Y, double& Z)
{
    double sl, cl, so, co;
    sincos(lat * DEG2RAD, &sl, &cl);
    sincos(lon * DEG2RAD, &so, &co);
    const double a2 = WGS84_A * WGS84_A, b2 = WGS84_B * WGS84_B;
    const double Nn = a2 / sqrt(a2 * cl * cl + b2 * sl * sl);
    const double ba = WGS84_B / WGS84_A;
    X = (Nn + alt) * cl * co;
    Y = (Nn + alt) * cl * so;
    Z = (Nn * (ba * ba) + alt) * sl;
}

// sphharmlag.py:345-359: Rodrigues rotation about k = (kx, ky, 0) by +theta0 (sign as written, F3)
__device__ __forceinline__ Geom sph_geom(const SphDev& M, double lat, double lon, double alt)
{
    Geom g;
    geodetic2ecef(lat, lon, alt, g.X, g.Y, g.Z);
    const double kd = M.kx * g.X + M.ky * g.Y;
    const double omc = 1.0 - M.rc;
    const double Rx = g.X * M.rc + (M.ky * g.Z) * M.rs + M.kx * kd * omc;
    const double Ry = g.Y * M.rc + (-M.kx * g.Z) * M.rs + M.ky * kd * omc;
    const double Rz = g.Z * M.rc + (M.kx * g.Y - M.ky * g.X) * M.rs;
    const double rho2 = Rx * Rx + Ry * Ry;
    const double r = sqrt(rho2 + Rz * Rz);
    g.x = Rz / r;
    g.s = sqrt(1.0 - g.x * g.x);          // scipy's lpmv forms (1-x^2)^(m/2) from x
    const double rho = sqrt(rho2);
    const bool pole = !(rho > 0.0);
    g.cphi = pole ? 1.0 : Rx / rho;       // arctan2(0,0) = 0
    g.sphi = pole ? 0.0 : Ry / rho;
    g.z = 100.0 * (r / M.RE - 1.0);
    g.Rx = Rx;
    g.Ry = Ry;
    return g;
}

// 2F1(a,b;c;zz) series with host-tabulated term ratios q[i] = (a+i)(b+i)/((c+i)(i+1)); the exit test
// is wave-uniform so the table stays on the scalar path.
__device__ __forceinline__ double hyp_series(const double* __restrict__ q, int nterms, double zz)
{
    double r = 1.0, sum = 1.0;
    // four terms per exit test: the ratios of a block arrive in one wide scalar load instead of four dependent ones
    // (measured at MAXL = 12, CAP_LIM = 15: the 24 seed series were ~40 % of the evaluation kernel); the up to
    // three terms added past convergence are below 1e-17 of the sum
    int i = 0;
    for (; i + 3 < nterms; i += 4) {
        const double q0 = q[i], q1 = q[i + 1], q2 = q[i + 2], q3 = q[i + 3];
        r *= q0 * zz;
        sum += r;
        r *= q1 * zz;
        sum += r;
        r *= q2 * zz;
        sum += r;
        r *= q3 * zz;
        sum += r;
        if (__all(fabs(r) <= 1e-17 * fabs(sum))) return sum;
    }
    for (; i < nterms; ++i) {
        r *= q[i] * zz;
        sum += r;
    }
    return sum;
}

// scipy.special.eval_laguerre(k, z), k = 0..maxk-1, by the three-term recurrence (sphharmlag.py:141)
template <int KCAP>
__device__ __forceinline__ void laguerre(int maxk, double z, double* Lk)
{
    Lk[0] = 1.0;
    if (KCAP > 1) Lk[1] = 1.0 - z;
#pragma unroll
    for (int k = 1; k + 1 < KCAP; ++k) {
        const double inv = 1.0 / (double)(k + 1);
        Lk[k + 1] = ((2.0 * k + 1.0 - z) * Lk[k] - (double)k * Lk[k - 1]) * inv;
    }
    (void)maxk;
}

}  // namespace
