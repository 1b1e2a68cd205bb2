// 10^x in plain IEEE double operations, the same on the host and on the device.
//
// The root finder of the regularisation-parameter search works in x = log10(alpha) (interpolate.py:214-216) and every
// iterate needs alpha = 10^x for X(alpha) = A^T W A + alpha R.  The iteration runs on the host (alpha_search.BrentBatch) or
// inside one kernel (k_brent_warm); for the two to see the same numbers bit for bit they must form alpha the same way, and
// the math libraries of the two sides do not promise that.  This routine uses only +, -, *, explicit fma, rint and ldexp
// (all correctly rounded / exact), with multiply-add contraction switched off, so it is one function on both sides.
// Accuracy ~1 ulp (x log2(10) is carried in two doubles); valid for |x| < 300.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define VI_HD __host__ __device__
#else
#define VI_HD
#endif

VI_HD inline double vi_exp10(double x)
{
#pragma clang fp contract(off)
    // log2(10) = L_HI + L_LO
    const double L_HI = 3.3219280948873622, L_LO = 1.6616175169735920e-16;
    const double th = x * L_HI;
    const double tl = fma(x, L_HI, -th) + x * L_LO;            // exact error of the product + the low part
    const double n = rint(th);
    const double f = (th - n) + tl;                             // |f| <= 0.5 (+ tiny)
    // 2^f = e^(f ln 2), ln 2 = G_HI + G_LO
    const double G_HI = 0.69314718055994529, G_LO = 2.3190468138462996e-17;
    const double rh = f * G_HI;
    const double rl = fma(f, G_HI, -rh) + f * G_LO;
    const double r = rh + rl;                                   // |r| <= 0.347
    // e^r by its Taylor series to r^17 / 17! (< 2e-23 relative), Horner from the top
    double p = 1.0 / 355687428096000.0;
    p = p * r + 1.0 / 20922789888000.0;
    p = p * r + 1.0 / 1307674368000.0;
    p = p * r + 1.0 / 87178291200.0;
    p = p * r + 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    return ldexp(p, (int)n);
}
