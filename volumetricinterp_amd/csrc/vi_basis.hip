// Basis-function kernels for gfx950: geodetic -> model coordinates, Laguerre x spherical-cap
// harmonics (reference: volumetricinterp/models/sphharmlag.py:118-145, :324-359) and Gaussian RBF
// (reference: volumetricinterp/models/radbasfun.py:83-112), as
//   K1  k_basis_*  : materialise A (used by the fit),
//   K2  k_eval_*   : fused evaluation  out[t,q] = sum_n A[q,n] C[t,n]  (Estimate.__call__,
//                    estimate.py:110-123) that never writes A, with the convex-hull test fused in.
// One thread per point; lat/lon/alt are read as coalesced SoA streams; every table the 64 lanes of
// a wave share (recurrence coefficients, coefficient tiles) is addressed wave-uniformly so it is
// served by the scalar data path / LDS broadcast and never costs per-lane HBM traffic.
#include "vi_common.h"
#include "vi_sph_device.h"

#include <cstdlib>

namespace {

// The shared per-point engine.  Sink::consume(l, cur[], cm[], sm[]) is called once per degree l with
// cur[m] = (normalised) P_{nu_l}^m(cos theta), m = 0..l.
template <int LCAP, int KCAP, class Sink>
__device__ __forceinline__ void sph_point(const SphDev& M, const Geom& g, Sink& sink)
{
    const int maxl = M.maxl;
    // azimuthal factors cos(m phi), sin(m phi) by angle addition (sphharmlag.py:278-281 takes them of |m| phi)
    double cm[LCAP], sm[LCAP];
    cm[0] = 1.0;
    sm[0] = 0.0;
#pragma unroll
    for (int m = 1; m < LCAP; ++m) {
        cm[m] = cm[m - 1] * g.cphi - sm[m - 1] * g.sphi;
        sm[m] = sm[m - 1] * g.cphi + cm[m - 1] * g.sphi;
    }
    const double x = g.x;
    const double zz = 0.5 * (1.0 - x);
    const int ng = M.ngroups;
    for (int gi = 0; gi < ng; ++gi) {
        const SphGroupDev G = M.groups[gi];
        const bool intseed = (G.nterms == 0);
        double cur[LCAP], prev[LCAP];
#pragma unroll
        for (int m = 0; m < LCAP; ++m) { cur[m] = 0.0; prev[m] = 0.0; }
        double pmm = 1.0;       // (-1)^m (2m-1)!! s^m
        double spow = 1.0;      // s^m
        const int nvmax = G.nvmax;
        for (int j = 0; j <= nvmax; ++j) {
            const double* __restrict__ cj = G.c + (size_t)j * maxl;
#pragma unroll
            for (int m = 0; m < LCAP; ++m) {
                if (m < maxl) {
                    if (j > m + 1) {
                        const double nw = fma(x, cur[m], -(cj[m] * prev[m]));
                        prev[m] = cur[m];
                        cur[m] = nw;
                    } else if (j == m) {
                        if (m > 0) { pmm *= -(2.0 * m - 1.0) * g.s; spow *= g.s; }
                        if (intseed) cur[m] = pmm;
                        else cur[m] = G.pref[m] * spow * hyp_series(G.q + (size_t)m * G.nterms, G.nterms, zz);
                    } else if (j == m + 1) {
                        prev[m] = cur[m];
                        if (intseed) cur[m] = x * (2.0 * m + 1.0) * cur[m];
                        else cur[m] = G.pref[maxl + m] * spow *
                                      hyp_series(G.q + (size_t)(maxl + m) * G.nterms, G.nterms, zz);
                    }
                }
            }
            const int l = G.pick[j];
            if (l >= 0) sink.template consume<LCAP>(l, cur, cm, sm);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K1: basis assembly.  A[p*ld_p + n*ld_n], n = k*maxl^2 + l(l+1) + m  (sphharmlag.py:79-99)
template <int KCAP>
struct BasisSink {
    double* A;
    int64_t ld_n;
    int maxk, L2;
    const double* __restrict__ scale;
    double ELk[KCAP];
    bool active;
    template <int LCAP>
    __device__ __forceinline__ void consume(int l, const double* cur, const double* cm, const double* sm)
    {
        const int r0 = l * (l + 1);
#pragma unroll
        for (int m = 0; m < LCAP; ++m) {
            if (m <= l) {
                const double fp = scale[r0 + m] * cm[m] * cur[m];
                const double fm = scale[r0 - m] * sm[m] * cur[m];
#pragma unroll
                for (int k = 0; k < KCAP; ++k) {
                    if (k < maxk && active) {
                        A[(int64_t)(k * L2 + r0 + m) * ld_n] = ELk[k] * fp;
                        if (m > 0) A[(int64_t)(k * L2 + r0 - m) * ld_n] = ELk[k] * fm;
                    }
                }
            }
        }
    }
};

template <int LCAP, int KCAP>
__global__ __launch_bounds__(BLOCK) void k_basis_sph(SphDev M, int64_t P, const double* __restrict__ lat,
                                                     const double* __restrict__ lon, const double* __restrict__ alt,
                                                     double* __restrict__ A, int64_t ld_p, int64_t ld_n)
{
    const int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t pc = p < P ? p : P - 1;
    const Geom g = sph_geom(M, lat[pc], lon[pc], alt[pc]);
    BasisSink<KCAP> sink;
    sink.A = A + pc * ld_p;
    sink.ld_n = ld_n;
    sink.maxk = M.maxk;
    sink.L2 = M.maxl * M.maxl;
    sink.scale = M.scale;
    sink.active = p < P;
    double Lk[KCAP];
    laguerre<KCAP>(M.maxk, g.z, Lk);
    const double E = exp(-0.5 * g.z);
#pragma unroll
    for (int k = 0; k < KCAP; ++k) sink.ELk[k] = E * Lk[k];
    sph_point<LCAP, KCAP>(M, g, sink);
}

// ---------------------------------------------------------------------------------------------
// Gradient basis (sphharmlag.py:148-184): for every basis function the components along z, theta, phi,
//   zhat = -e/2 (L0 + 2 L1) Pmv A 100/RE,  that = e L0 (-(v+1) x Pmv + (v-m+1) Pmv1) A / (y (z/100+1) RE),
//   phat = e L0 Pmv dAz / (y (z/100+1) RE),
// with L1 = eval_genlaguerre(k-1, 1, z), Pmv1 = lpmv(m, v+1, x).  Same chains as the basis, run one degree
// further: when the chain has reached degree nu_l + 1, prev holds P at nu_l and cur at nu_l + 1.
// CONTRACT = false: store the (P, 3, N) gradient basis.  CONTRACT = true: contract it on the fly with one coefficient
// vector Cv (N) and store only the three gradient components per point (ld_p = 3, ld_c = 1): the gradient of the fitted
// parameter, never materialising 3N values per point.
template <int LCAP, int KCAP, bool CONTRACT>
__global__ __launch_bounds__(BLOCK) void k_grad_sph(SphDev M, int64_t P, const double* __restrict__ lat,
                                                    const double* __restrict__ lon, const double* __restrict__ alt,
                                                    const double* __restrict__ Cv, double* __restrict__ Gout,
                                                    int64_t ld_p, int64_t ld_c, int64_t ld_n)
{
    double az = 0.0, at = 0.0, ap = 0.0;
    const int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t pc = p < P ? p : P - 1;
    const bool active = p < P;
    const Geom g = sph_geom(M, lat[pc], lon[pc], alt[pc]);
    double* Gp = Gout + pc * ld_p;
    const int maxl = M.maxl, maxk = M.maxk, L2 = maxl * maxl;
    double L0[KCAP], L1[KCAP];          // L_k(z) and L^(1)_{k-1}(z)
    laguerre<KCAP>(maxk, g.z, L0);
    L1[0] = 0.0;                        // eval_genlaguerre(-1, 1, z) = 0
    if (KCAP > 1) L1[1] = 1.0;
    if (KCAP > 2) L1[2] = 2.0 - g.z;
#pragma unroll
    for (int k = 3; k < KCAP; ++k) {    // n L^(1)_n = (2n - z) L^(1)_{n-1} - n L^(1)_{n-2},  n = k-1
        const int n = k - 1;
        L1[k] = ((2.0 * n - g.z) * L1[k - 1] - (double)n * L1[k - 2]) / (double)n;
    }
    const double e = exp(-0.5 * g.z);
    const double x = g.x, y = g.s;
    const double inv_hr = 1.0 / (y * (g.z / 100.0 + 1.0) * M.RE);       // 1 / (y (z/100+1) RE)
    const double zfac = -0.5 * e * 100.0 / M.RE;
    double cm[LCAP], sm[LCAP];
    cm[0] = 1.0;
    sm[0] = 0.0;
#pragma unroll
    for (int m = 1; m < LCAP; ++m) {
        cm[m] = cm[m - 1] * g.cphi - sm[m - 1] * g.sphi;
        sm[m] = sm[m - 1] * g.cphi + cm[m - 1] * g.sphi;
    }
    const double zz = 0.5 * (1.0 - x);
    for (int gi = 0; gi < M.ngroups; ++gi) {
        const SphGroupDev G = M.groups[gi];
        const bool intseed = (G.nterms == 0);
        double cur[LCAP], prev[LCAP];
#pragma unroll
        for (int m = 0; m < LCAP; ++m) { cur[m] = 0.0; prev[m] = 0.0; }
        double pmm = 1.0, spow = 1.0;
        for (int j = 0; j <= G.nvmax + 1; ++j) {
            const double* __restrict__ cj = G.c + (size_t)j * maxl;
#pragma unroll
            for (int m = 0; m < LCAP; ++m) {
                if (m < maxl) {
                    if (j > m + 1) {
                        const double nw = fma(x, cur[m], -(cj[m] * prev[m]));
                        prev[m] = cur[m];
                        cur[m] = nw;
                    } else if (j == m) {
                        if (m > 0) { pmm *= -(2.0 * m - 1.0) * g.s; spow *= g.s; }
                        if (intseed) cur[m] = pmm;
                        else cur[m] = G.pref[m] * spow * hyp_series(G.q + (size_t)m * G.nterms, G.nterms, zz);
                    } else if (j == m + 1) {
                        prev[m] = cur[m];
                        if (intseed) cur[m] = x * (2.0 * m + 1.0) * cur[m];
                        else cur[m] = G.pref[maxl + m] * spow *
                                      hyp_series(G.q + (size_t)(maxl + m) * G.nterms, G.nterms, zz);
                    }
                }
            }
            const int l = j >= 1 ? G.pick[j - 1] : -1;      // degree nu_l + 1 reached
            if (l >= 0) {
                const int r0 = l * (l + 1);
                const double v = M.nu[l];
#pragma unroll
                for (int m = 0; m < LCAP; ++m) {
                    if (m <= l) {
#pragma unroll
                        for (int sgn = 0; sgn < 2; ++sgn) {
                            if (sgn == 1 && m == 0) continue;
                            const int r = sgn ? r0 - m : r0 + m;
                            const double ms = sgn ? -(double)m : (double)m;            // signed order
                            const double Q0 = M.scale[r] * prev[m];                      // Kvm * lpmv(m, v, x)
                            const double Q1 = M.scale1[r] * cur[m];                      // Kvm * lpmv(m, v+1, x)
                            const double trig = sgn ? sm[m] : cm[m];
                            const double dtrig = sgn ? (double)m * cm[m] : -(double)m * sm[m];
                            const double tz = zfac * Q0 * trig;
                            const double tt = e * (-(v + 1.0) * x * Q0 + (v - ms + 1.0) * Q1) * trig * inv_hr;
                            const double tp = e * Q0 * dtrig * inv_hr;
#pragma unroll
                            for (int k = 0; k < KCAP; ++k) {
                                if (CONTRACT) {
                                    if (k < maxk) {
                                        const double cf = Cv[k * L2 + r];
                                        az = fma(tz * (L0[k] + 2.0 * L1[k]), cf, az);
                                        at = fma(tt * L0[k], cf, at);
                                        ap = fma(tp * L0[k], cf, ap);
                                    }
                                } else if (k < maxk && active) {
                                    double* o = Gp + (int64_t)(k * L2 + r) * ld_n;
                                    o[0] = tz * (L0[k] + 2.0 * L1[k]);
                                    o[ld_c] = tt * L0[k];
                                    o[2 * ld_c] = tp * L0[k];
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (CONTRACT && active) {
        Gp[0] = az;
        Gp[ld_c] = at;
        Gp[2 * ld_c] = ap;
    }
}

// ---------------------------------------------------------------------------------------------
// K2: fused evaluation for TT timesteps per pass.  Cp is the coefficient tile reordered to
// [t][r = l(l+1)+m][k] and pre-multiplied by the per-(l,m) constant (see k_prep_coef).
template <int KCAP, int TT>
struct EvalSink {
    const double* __restrict__ Cp;   // [TT][L2*maxk], wave-uniform reads
    int maxk, NB;
    double Lk[KCAP];
    double acc[TT];
    template <int LCAP>
    __device__ __forceinline__ void consume(int l, const double* cur, const double* cm, const double* sm)
    {
        const int r0 = l * (l + 1);
#pragma unroll
        for (int m = 0; m < LCAP; ++m) {
            if (m <= l) {
                const double* __restrict__ cp = Cp + (size_t)(r0 + m) * maxk;
                const double* __restrict__ cn = Cp + (size_t)(r0 - m) * maxk;
                const double pc = cur[m] * cm[m];
                const double ps = cur[m] * sm[m];
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    double Sp = 0.0, Sm = 0.0;
#pragma unroll
                    for (int k = 0; k < KCAP; ++k) {
                        if (k < maxk) {
                            Sp = fma(cp[(size_t)t * NB + k], Lk[k], Sp);
                            if (m > 0) Sm = fma(cn[(size_t)t * NB + k], Lk[k], Sm);
                        }
                    }
                    acc[t] = fma(pc, Sp, acc[t]);
                    if (m > 0) acc[t] = fma(ps, Sm, acc[t]);
                }
            }
        }
    }
};

// Convex-hull test (estimate.py:153-178 semantics: inside <=> max_f (n_f . x + d_f) <= tol), as a mask pass.
// `hull` is the internal buffer built by k_prep_hull: [c0 (3 doubles, 1 pad)] [F x 4 fp64 facet equations]
// [F x float4 facets relative to c0].  With ~460 facets an fp64 test inside the evaluation kernel cost twice the
// whole basis (dependent scalar loads), so the test is its own pass: the fp32 facets (wave-uniform scalar loads, see
// k_hull_mask) are evaluated relative to a reference point c0 on the plane of facet 0; only points within a band of the
// surface repeat the test in fp64.  The band covers the fp32 error of the prefilter: the float casts of normal and
// offset plus three FMA roundings are a few units of 2^-24 |x - c0|, and c0 - the foot of the origin's perpendicular
// on a facet plane - may lie far from the hull when that plane passes near the Earth's centre, so the band grows with
// the distance: HULL_BAND + HULL_BAND_REL |x - c0| (4 m at the reference point, ~10 m at 6000 km).  Points with
// non-finite coordinates are outside.  The byte mask is reused by every timestep tile of the evaluation.
constexpr float HULL_BAND = 4.0f;
constexpr float HULL_BAND_REL = 1.0e-6f;
constexpr int HULL_HDR = 8;          // doubles in front of the facet equations: c0 (3), s, offmax, 3 spare

constexpr int HULL_PP = 8;          // points per thread: every facet fetched serves eight points
constexpr int HULL_PAD = 16;        // the fp32 facet list is padded to a multiple of 16 with repeats of facet 0

// Round 3: the facets come through wave-uniform (scalar) loads - four planes per s_load_dwordx16, used as SGPR operands
// of the packed FMAs - instead of LDS broadcasts: at 4 points per thread the pass issued one ds_read_b128 per facet and
// thread, 8 LDS clocks for every 46 VALU clocks of a SIMD with four SIMDs on one LDS pipe (~70 % busy), plus a v_mov
// per facet; eight points per thread halve what is left per point (early-exit test, plane set-up).
__global__ __launch_bounds__(BLOCK, 4) void k_hull_mask(int64_t Q, const double* __restrict__ lat,
                                                     const double* __restrict__ lon, const double* __restrict__ alt,
                                                     const double* __restrict__ hull, int F, double tol,
                                                     unsigned char* __restrict__ mask)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float4* __restrict__ pl = reinterpret_cast<const float4*>(hull + HULL_HDR + 4 * (size_t)F);
    const int Fp = (F + HULL_PAD - 1) / HULL_PAD * HULL_PAD;
    const int64_t q0 = (int64_t)blockIdx.x * (BLOCK * HULL_PP) + threadIdx.x;      // points q0 + u * BLOCK
    const double c0x = hull[0], c0y = hull[1], c0z = hull[2];
    // geodetic -> ECEF one point after the other (one copy of the fp64 trigonometry, few registers), the fp32 offsets
    // from c0 handed to the facet loop's registers through LDS (each thread reads back what it wrote: no barrier)
    __shared__ float sh_d[HULL_PP][3][BLOCK];
#pragma unroll 1
    for (int u = 0; u < HULL_PP; ++u) {
        const int64_t q = q0 + (int64_t)u * BLOCK;
        const int64_t qc = q < Q ? q : Q - 1;
        double X, Y, Z;
        geodetic2ecef(lat[qc], lon[qc], alt[qc], X, Y, Z);
        sh_d[u][0][threadIdx.x] = (float)(X - c0x);
        sh_d[u][1][threadIdx.x] = (float)(Y - c0y);
        sh_d[u][2][threadIdx.x] = (float)(Z - c0z);
    }
    f2 px[HULL_PP / 2], py[HULL_PP / 2], pz[HULL_PP / 2], dmax[HULL_PP / 2], thr[HULL_PP / 2];
    bool finite[HULL_PP];
#pragma unroll
    for (int u = 0; u < HULL_PP; ++u) {
        const float dx = sh_d[u][0][threadIdx.x], dy = sh_d[u][1][threadIdx.x], dz = sh_d[u][2][threadIdx.x];
        finite[u] = fabsf(dx) + fabsf(dy) + fabsf(dz) < 3.0e38f;                   // false for NaN / inf coordinates
        px[u / 2][u % 2] = dx;
        py[u / 2][u % 2] = dy;
        pz[u / 2][u % 2] = dz;
        dmax[u / 2][u % 2] = -3.0e38f;
        thr[u / 2][u % 2] = HULL_BAND + HULL_BAND_REL * sqrtf(dx * dx + dy * dy + dz * dz);       // the band of the point
    }
    auto dist = [&](const float4 p, int h) {
        return __builtin_elementwise_fma(f2{p.x, p.x}, px[h],
                                         __builtin_elementwise_fma(f2{p.y, p.y}, py[h],
                                                                   __builtin_elementwise_fma(f2{p.z, p.z}, pz[h], f2{p.w, p.w})));
    };
    // four facets per round, the next four fetched while these are applied (the list carries one spare group)
    float4 n0 = pl[0], n1 = pl[1], n2 = pl[2], n3 = pl[3];
#pragma unroll 1
    for (int f = 0; f < Fp; f += 4) {
        const float4 p0 = n0, p1 = n1, p2 = n2, p3 = n3;
        n0 = pl[f + 4]; n1 = pl[f + 5]; n2 = pl[f + 6]; n3 = pl[f + 7];
#pragma unroll
        for (int h = 0; h < HULL_PP / 2; ++h) {
            const f2 d0 = dist(p0, h), d1 = dist(p1, h), d2 = dist(p2, h), d3 = dist(p3, h);
            dmax[h].x = fmaxf(fmaxf(dmax[h].x, d0.x), d1.x);
            dmax[h].y = fmaxf(fmaxf(dmax[h].y, d0.y), d1.y);
            dmax[h].x = fmaxf(fmaxf(dmax[h].x, d2.x), d3.x);
            dmax[h].y = fmaxf(fmaxf(dmax[h].y, d2.y), d3.y);
        }
        if ((f & (HULL_PAD - 4)) == HULL_PAD - 4) {
            // a point is outside as soon as ONE facet says so: leave when every point of the wave is decided
            float slack = 3.0e38f;
#pragma unroll
            for (int h = 0; h < HULL_PP / 2; ++h)
                slack = fminf(slack, fminf(dmax[h].x - thr[h].x, dmax[h].y - thr[h].y));
            if (__all(slack > (float)tol)) break;
        }
    }
    unsigned inbits = 0, border = 0;
#pragma unroll
    for (int u = 0; u < HULL_PP; ++u) {
        const float dm = dmax[u / 2][u % 2], bd = thr[u / 2][u % 2];
        if (!finite[u] || dm > (float)tol + bd) continue;                // outside
        if (dm < (float)tol - bd) inbits |= 1u << u;                      // inside
        else border |= 1u << u;                                           // within the band of the surface
    }
    // Points within the band: the exact fp64 test, one point at a time with the facets spread over the 64 lanes of the wave
    // (a lane running all F facets alone - F dependent scalar loads - held its whole wave for longer than the fp32 pass of
    // the entire grid takes: 150 us measured against 95 us with every point deep inside the hull)
    const int lane = threadIdx.x & 63;
    const double* __restrict__ eq = hull + HULL_HDR;
#pragma unroll 1
    for (int u = 0; u < HULL_PP; ++u) {
        const int64_t q = q0 + (int64_t)u * BLOCK;
        const bool bl = ((border >> u) & 1u) && q < Q;
        bool in = (inbits >> u) & 1u;
        unsigned long long todo = __ballot(bl);
        if (todo) {
            double X = 0.0, Y = 0.0, Z = 0.0;
            if (bl) geodetic2ecef(lat[q], lon[q], alt[q], X, Y, Z);
            while (todo) {
                const int src = __ffsll(todo) - 1;
                todo &= todo - 1;
                const double xs = __shfl(X, src), ys = __shfl(Y, src), zs = __shfl(Z, src);
                bool viol = false;
                for (int g = lane; g < F; g += 64) {
                    const double d = fma(eq[4 * g], xs, fma(eq[4 * g + 1], ys, fma(eq[4 * g + 2], zs, eq[4 * g + 3])));
                    viol = viol || !(d <= tol);
                }
                const bool any = __any(viol);
                if (lane == src) in = !any;
            }
        }
        if (q < Q) mask[q] = in ? 1 : 0;
    }
}

// ---- the hull pass on the matrix cores (round 4) -------------------------------------------------------------------------
// The plane distances of 32 points from 32 facets are ONE v_mfma_f32_32x32x16_f16: every coordinate (relative to c0, in units
// of HULL_UNIT metres) and every normal component (times HULL_NSCALE) is split into two halves hi + lo (22 bits together), and
// the sixteen k-slots of the instruction hold the four cross products of the three components plus the plane offset:
//      k     0    1    2    3    4    5    6    7  |  8    9    10   11   12    13    14 15
//  A  facet  nxh  nyh  nzh  nxl  nyl  nzl  nxh  nyh |  nzh  nxl  nyl  nzl  offh  offl  0  0
//  B  point  xh   yh   zh   xh   yh   zh   xl   yl  |  zl   xl   yl   zl   1024  1024  0  0
// (lanes 0-31 carry k 0-7 of row / column lane, lanes 32-63 k 8-15: cdna4 32x32x16 operand map).  The products of two halves
// are exact in fp32; what is lost is the split (2^-22 of |x|, |n|, |off| each), the flush of a subnormal lo half (6e-8 |x|)
// and the fp32 accumulation of sixteen terms (<= 16 x 2^-23 of (|x - c0| + |off|)): together < 2.5e-6 (|x - c0| + max |off|),
// covered by the band HULL_BAND + HULL_MX_REL (|x - c0| + max |off|), inside which a point repeats the test in fp64 - the
// mask is the fp64 definition's whatever the prefilter does inside its band.  The result column is the lane's point and the
// sixteen registers are sixteen facets, so the maximum over the facets is eight v_max3_f32 per tile and lane and one
// v_permlane32_swap at the end.  Measured (tools/microbench/mfma_max3_overlap.hip): 53 cycles per 1024 distances and SIMD with
// zero operands, 74 with real ones - the shader clock drops from 2.3 to 1.6 GHz under the load, the loop is power-bound -
// against ~190 of the packed-fp32 loop (k_hull_mask); 128^3 points x 460 facets: 88 -> 46 us over the call without a hull.
// Normals longer than 1 are scaled down by s = max |n| (Qhull's are unit vectors: s = 1), points further than HULL_FAR from c0
// and lists with a non-finite entry or an offset beyond the fp16 range go to the fp64 test as they are.
constexpr float HULL_UNIT = 256.0f;          // metres per unit of the fp16 coordinates: 65504 units = 16 769 km
constexpr float HULL_NSCALE = 1024.0f;       // the unit normal's components times this: the lo half stays a normal fp16 number
constexpr float HULL_FAR = 1.6e7f;           // metres from c0 beyond which a point skips the fp16 prefilter
constexpr float HULL_MX_REL = 4.0e-6f;
constexpr int HULL_MX_PP = 4;                // points per thread = eight 32-point groups per wave and facet tile

__device__ __forceinline__ unsigned hull_split(float v)          // (hi, lo) halves of v packed: hi in bits 0-15
{
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    return (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
}

// geodetic2ecef with sincos_cw (vi_sph_device.h); false when an angle is outside the range of its reduction (or NaN)
__device__ __forceinline__ bool hull_geodetic2ecef(double lat, double lon, double alt, double& X, double& Y, double& Z)
{
    const double la = lat * DEG2RAD, lo = lon * DEG2RAD;
    double sl, cl, so, co;
    sincos_cw(la, sl, cl);
    sincos_cw(lo, so, co);
    const double a2 = WGS84_A * WGS84_A, b2 = WGS84_B * WGS84_B;
    const double Nn = a2 / sqrt(a2 * cl * cl + b2 * sl * sl);
    const double ba = WGS84_B / WGS84_A;
    X = (Nn + alt) * cl * co;
    Y = (Nn + alt) * cl * so;
    Z = (Nn * (ba * ba) + alt) * sl;
    return fabs(la) < 1.0e4 && fabs(lo) < 1.0e4;
}

typedef float hull_f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float hull_max16(const hull_f16v& c, float d)       // max(d, the sixteen results): eight v_max3_f32
{
    const float t0 = fmaxf(fmaxf(c[0], c[1]), c[2]), t1 = fmaxf(fmaxf(c[3], c[4]), c[5]), t2 = fmaxf(fmaxf(c[6], c[7]), c[8]),
                t3 = fmaxf(fmaxf(c[9], c[10]), c[11]), t4 = fmaxf(fmaxf(c[12], c[13]), c[14]), t5 = fmaxf(fmaxf(c[15], d), t0),
                t6 = fmaxf(fmaxf(t1, t2), t3);
    return fmaxf(fmaxf(t4, t5), t6);
}

template <int PP>
__global__ __launch_bounds__(BLOCK, 4) void k_hull_mask_mx(int64_t Q, const double* __restrict__ lat,
                                                           const double* __restrict__ lon, const double* __restrict__ alt,
                                                           const double* __restrict__ hull, int F, double tol,
                                                           unsigned char* __restrict__ mask)
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f16v __attribute__((ext_vector_type(16)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int Fp = (F + HULL_PAD - 1) / HULL_PAD * HULL_PAD;
    const int T = (F + 31) / 32;
    const u4* __restrict__ tiles = reinterpret_cast<const u4*>(hull + HULL_HDR + 4 * (size_t)F) + (Fp + 4);
    const int lane = threadIdx.x & 63;
    const int64_t q0 = (int64_t)blockIdx.x * (BLOCK * PP) + threadIdx.x;           // points q0 + u * BLOCK
    const double c0x = hull[0], c0y = hull[1], c0z = hull[2];
    const float s = (float)hull[3];
    const float offmax = (float)hull[4];                                         // max |n . c0 + d| / s; inf: no prefilter
    // geodetic -> ECEF of the thread's PP points, unrolled: their 3 PP loads are in flight together (one after the other a
    // wave waited out a memory latency per point: the pass reads 24 B a point, 50 MB for 128^3).  thr < 0 marks a point the
    // prefilter does not judge (non-finite, far away, angle beyond the range of sincos_cw).
    // The B operands: lanes 0-31 hold k 0-7 of their column, lanes 32-63 k 8-15; one swap of the upper half of one register
    // with the lower half of another makes a register of each of the two groups a 64-lane set of points consists of.
    unsigned Bf[2 * PP][4];
    float thr[PP];
    double qlat[PP], qlon[PP], qalt[PP];
#pragma unroll
    for (int u = 0; u < PP; ++u) {
        const int64_t q = q0 + (int64_t)u * BLOCK;
        const int64_t qc = q < Q ? q : Q - 1;
        qlat[u] = lat[qc];
        qlon[u] = lon[qc];
        qalt[u] = alt[qc];
    }
#pragma unroll
    for (int u = 0; u < PP; ++u) {
        double X, Y, Z;
        const bool inrange = hull_geodetic2ecef(qlat[u], qlon[u], qalt[u], X, Y, Z);
        float vx = (float)((X - c0x) * (1.0 / HULL_UNIT)), vy = (float)((Y - c0y) * (1.0 / HULL_UNIT)),
              vz = (float)((Z - c0z) * (1.0 / HULL_UNIT));
        const float r = HULL_UNIT * sqrtf(vx * vx + vy * vy + vz * vz);
        const bool ok = inrange && r < HULL_FAR;                                 // false for NaN / inf coordinates too
        if (!ok) vx = vy = vz = 0.0f;
        const unsigned sx = hull_split(vx), sy = hull_split(vy), sz = hull_split(vz);
        thr[u] = ok ? s * (HULL_BAND + HULL_MX_REL * (r + offmax)) : -1.0f;
        const unsigned p0 = (sx & 0xffffu) | (sy << 16);                         // (xh, yh)
        const unsigned p1 = (sz & 0xffffu) | (sx & 0xffff0000u);                 // (zh, xl)
        const unsigned p2 = (sy >> 16) | (sz & 0xffff0000u);                     // (yl, zl)
        const unsigned lo[4] = {p0,                                              // (xh, yh)
                                (p1 & 0xffffu) | (p0 << 16),                     // (zh, xh)
                                (p0 >> 16) | (p1 << 16),                         // (yh, zh)
                                (p1 >> 16) | (p2 << 16)};                        // (xl, yl)
        const unsigned hi[4] = {(p2 >> 16) | (p1 & 0xffff0000u),                 // (zl, xl)
                                p2,                                              // (yl, zl)
                                0x64006400u,                                     // (1024, 1024)
                                0u};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const auto w = __builtin_amdgcn_permlane32_swap(lo[j], hi[j], false, false);
            Bf[2 * u][j] = w[0];                    // lanes 0-31: lo of their own point; 32-63: hi of the point of lane - 32
            Bf[2 * u + 1][j] = w[1];                // lanes 0-31: lo of the point of lane + 32; 32-63: hi of their own
        }
    }
    float dmax[2 * PP];
#pragma unroll
    for (int g = 0; g < 2 * PP; ++g) dmax[g] = -3.0e38f;
    const float cs = s * (HULL_UNIT / HULL_NSCALE);                              // accumulator units -> the equations' units
    const float tolf = (float)tol;
    u4 An = tiles[lane];
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const u4 A = An;
        An = tiles[(size_t)(t + 1 < T ? t + 1 : t) * 64 + lane];
        // Two accumulators by hand: the product of group g + 1 is issued before the maxima of group g are taken.  Left to the
        // compiler the loop is product -> wait -> eight maxima -> next product on ONE accumulator, 85 cycles per product
        // and SIMD at four waves against 53 this way (tools/microbench/mfma_max3_overlap.hip).  The products are inline
        // assembly, so the wait between a matrix-core write and a vector read of it is ours to keep: s_nop 15 (16 wait
        // states; the 8-pass product needs 12) inside the same statement, so that nothing the compiler may place behind
        // it - the maxima, a copy - reads the accumulator early; "=&v": the result does not share registers with A or B.
        f16v pa, pb;
        // (`held`: the other accumulator, named as an operand so that its maxima stay behind this product - the compiler is
        // otherwise free to take them first and to put both accumulators into the same registers)
#define VI_HULL_MFMA(acc, g, held)                                                                                        \
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, 0\n\ts_nop 15"                                                      \
                 : "=&v"(acc), "+v"(held)                                                                                 \
                 : "v"(A), "v"((u4){Bf[g][0], Bf[g][1], Bf[g][2], Bf[g][3]}))
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\ts_nop 15"
                     : "=&v"(pa) : "v"(A), "v"((u4){Bf[0][0], Bf[0][1], Bf[0][2], Bf[0][3]}));
#pragma unroll
        for (int g = 0; g < 2 * PP; g += 2) {
            VI_HULL_MFMA(pb, g + 1, pa);
            dmax[g] = hull_max16(pa, dmax[g]);
            if (g + 2 < 2 * PP) VI_HULL_MFMA(pa, g + 2, pb);
            dmax[g + 1] = hull_max16(pb, dmax[g + 1]);
        }
#undef VI_HULL_MFMA
        if ((t & (t + 1)) == 0 && t + 1 < T) {
            // after tiles 0, 1, 3, 7, ...: a point is outside as soon as ONE facet says so - leave when every point of the
            // wave is decided (the host hands the facets over in greedy-cover order: the first tile decides 98 % of them)
            float slack = 3.0e38f;
#pragma unroll
            for (int u = 0; u < PP; ++u) {
                const auto w = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, dmax[2 * u]),
                                                                __builtin_bit_cast(unsigned, dmax[2 * u + 1]), false, false);
                const unsigned w0 = w[0], w1 = w[1];                  // (a bit_cast of w[1] itself reads w[0]: through scalars)
                const float dm = cs * fmaxf(__uint_as_float(w0), __uint_as_float(w1));
                slack = fminf(slack, thr[u] < 0.0f ? 3.0e38f : dm - thr[u]);
            }
            if (__all(slack > tolf)) break;
        }
    }
    unsigned inbits = 0, border = 0;
#pragma unroll
    for (int u = 0; u < PP; ++u) {
        // lanes 0-31: the point's two partial maxima are dmax[2u] of this lane and of lane + 32; lanes 32-63: dmax[2u + 1]
        const auto w = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, dmax[2 * u]),
                                                        __builtin_bit_cast(unsigned, dmax[2 * u + 1]), false, false);
        const unsigned w0 = w[0], w1 = w[1];                  // (a bit_cast of w[1] itself reads w[0]: through scalars)
        const float dm = cs * fmaxf(__uint_as_float(w0), __uint_as_float(w1));
        const float bd = thr[u];
        if (bd < 0.0f || !(fabsf(dm - tolf) > bd)) border |= 1u << u;             // not judged, or within the band (NaN too)
        else if (dm < tolf) inbits |= 1u << u;                                    // inside
    }
    // Points within the band: the exact fp64 test, one point at a time with the facets spread over the 64 lanes of the wave
    const double* __restrict__ eq = hull + HULL_HDR;
#pragma unroll 1
    for (int u = 0; u < PP; ++u) {
        const int64_t q = q0 + (int64_t)u * BLOCK;
        const bool bl = ((border >> u) & 1u) && q < Q;
        bool in = (inbits >> u) & 1u;
        unsigned long long todo = __ballot(bl);
        if (todo) {
            double X = 0.0, Y = 0.0, Z = 0.0;
            if (bl) geodetic2ecef(lat[q], lon[q], alt[q], X, Y, Z);
            while (todo) {
                const int src = __ffsll(todo) - 1;
                todo &= todo - 1;
                const double xs = __shfl(X, src), ys = __shfl(Y, src), zs = __shfl(Z, src);
                bool viol = false;
                for (int g = lane; g < F; g += 64) {
                    const double d = fma(eq[4 * g], xs, fma(eq[4 * g + 1], ys, fma(eq[4 * g + 2], zs, eq[4 * g + 3])));
                    viol = viol || !(d <= tol);
                }
                const bool any = __any(viol);
                if (lane == src) in = !any;
            }
        }
        if (q < Q) mask[q] = in ? 1 : 0;
    }
}

// hullbuf <- [c0, s, offmax][eq][float4 facets][fp16 operand tiles]; c0 = foot of the origin's perpendicular on facet 0 (a
// point of the hull surface), s = max(1, longest normal), offmax = max |n . c0 + d| / s (inf: a non-finite entry in the list or
// an offset beyond the fp16 range - every point then takes the fp64 test).  Every block forms s and offmax itself.
inline size_t hull_buf_bytes(size_t F)
{
    const size_t Fp = (F + HULL_PAD - 1) / HULL_PAD * HULL_PAD, T = (F + 31) / 32;
    return (HULL_HDR + 4 * F) * sizeof(double) + (Fp + 4) * sizeof(float4) + T * 64 * 16 + 64;
}

constexpr int HULL_PREP_BLOCKS = 4;
__device__ __forceinline__ void prep_hull_body(int F, const double* __restrict__ eq, double* __restrict__ hullbuf, int bid, int nb)
{
    const double c0x = -eq[3] * eq[0], c0y = -eq[3] * eq[1], c0z = -eq[3] * eq[2];
    __shared__ double sh_n[256], sh_o[256];
    double nm = 0.0, om = 0.0;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const double nx = eq[4 * f], ny = eq[4 * f + 1], nz = eq[4 * f + 2], off = nx * c0x + ny * c0y + nz * c0z + eq[4 * f + 3];
        const double n2 = nx * nx + ny * ny + nz * nz;
        nm = !(n2 <= 1.0e300) ? __builtin_inf() : fmax(nm, n2);
        om = !(fabs(off) <= 1.0e300) ? __builtin_inf() : fmax(om, fabs(off));
    }
    sh_n[threadIdx.x] = nm;
    sh_o[threadIdx.x] = om;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            sh_n[threadIdx.x] = fmax(sh_n[threadIdx.x], sh_n[threadIdx.x + w]);
            sh_o[threadIdx.x] = fmax(sh_o[threadIdx.x], sh_o[threadIdx.x + w]);
        }
        __syncthreads();
    }
    const double s = fmax(1.0, sqrt(sh_n[0]));
    double offmax = sh_o[0] / s;
    if (!(s <= 1.0e150) || !(offmax <= 65000.0 * HULL_UNIT)) offmax = __builtin_inf();
    if (threadIdx.x == 0 && bid == 0) {
        hullbuf[0] = c0x; hullbuf[1] = c0y; hullbuf[2] = c0z; hullbuf[3] = s;
        hullbuf[4] = offmax; hullbuf[5] = 0.0; hullbuf[6] = 0.0; hullbuf[7] = 0.0;
    }
    float4* pl = reinterpret_cast<float4*>(hullbuf + HULL_HDR + 4 * (size_t)F);
    for (int f = bid * blockDim.x + threadIdx.x; f < F; f += nb * blockDim.x) {
        const double nx = eq[4 * f], ny = eq[4 * f + 1], nz = eq[4 * f + 2], off = eq[4 * f + 3];
        double* e = hullbuf + HULL_HDR + 4 * (size_t)f;
        e[0] = nx; e[1] = ny; e[2] = nz; e[3] = off;
        pl[f] = make_float4((float)nx, (float)ny, (float)nz, (float)(nx * c0x + ny * c0y + nz * c0z + off));
    }
    // padding of the fp32 list (k_hull_mask reads it in groups of HULL_PAD): facet 0 again - the maximum does not change
    const int Fp = (F + HULL_PAD - 1) / HULL_PAD * HULL_PAD;
    for (int f = F + bid * blockDim.x + threadIdx.x; f < Fp + 4; f += nb * blockDim.x)      // + the spare group
        pl[f] = make_float4((float)eq[0], (float)eq[1], (float)eq[2], (float)(eq[0] * c0x + eq[1] * c0y + eq[2] * c0z + eq[3]));
    // the fp16 operand tiles of k_hull_mask_mx: tile t, lane l = facet 32 t + (l & 31) (facet 0 again past the end), k-slots
    // 8 (l >> 5) ... + 7 of the table above its kernel
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4* tiles = reinterpret_cast<u4*>(pl + Fp + 4);
    const int T = (F + 31) / 32;
    const bool usable = offmax < __builtin_inf();
    for (int i = bid * blockDim.x + threadIdx.x; i < T * 64; i += nb * blockDim.x) {
        int f = 32 * (i >> 6) + (i & 31);
        if (f >= F) f = 0;
        const double nx = eq[4 * f], ny = eq[4 * f + 1], nz = eq[4 * f + 2];
        const double off = nx * c0x + ny * c0y + nz * c0z + eq[4 * f + 3];
        u4 v = {0u, 0u, 0u, 0u};
        if (usable) {
            const unsigned sx = hull_split((float)(nx / s * HULL_NSCALE)), sy = hull_split((float)(ny / s * HULL_NSCALE)),
                           sz = hull_split((float)(nz / s * HULL_NSCALE)), so = hull_split((float)(off / s / HULL_UNIT));
            if (((i >> 5) & 1) == 0) {
                v[0] = (sx & 0xffffu) | (sy << 16);                  // (nxh, nyh)
                v[1] = (sz & 0xffffu) | (sx & 0xffff0000u);          // (nzh, nxl)
                v[2] = (sy >> 16) | (sz & 0xffff0000u);              // (nyl, nzl)
                v[3] = v[0];                                         // (nxh, nyh)
            } else {
                v[0] = (sz & 0xffffu) | (sx & 0xffff0000u);          // (nzh, nxl)
                v[1] = (sy >> 16) | (sz & 0xffff0000u);              // (nyl, nzl)
                v[2] = so;                                           // (offh, offl)
                v[3] = 0u;
            }
        }
        tiles[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_prep_hull(int F, const double* __restrict__ eq, double* __restrict__ hullbuf)
{
    prep_hull_body(F, eq, hullbuf, (int)blockIdx.x, (int)gridDim.x);
}

template <int LCAP, int KCAP, int TT>
__global__ __launch_bounds__(BLOCK) void k_eval_sph(SphDev M, int64_t Q, const double* __restrict__ lat,
                                                    const double* __restrict__ lon, const double* __restrict__ alt,
                                                    int tcount, const double* __restrict__ Cp,
                                                    const unsigned char* __restrict__ mask, int F, double tol,
                                                    double* __restrict__ out)
{
    const int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t qc = q < Q ? q : Q - 1;
    const Geom g = sph_geom(M, lat[qc], lon[qc], alt[qc]);
    bool in = true;
    if (F > 0) {
        in = mask[qc] != 0;
        if (!__any(in && q < Q)) {            // whole wave outside the hull: skip the basis work
            if (q < Q)
                for (int t = 0; t < tcount; ++t) out[(int64_t)t * Q + q] = __builtin_nan("");
            return;
        }
    }
    EvalSink<KCAP, TT> sink;
    sink.Cp = Cp;
    sink.maxk = M.maxk;
    sink.NB = M.N;
    laguerre<KCAP>(M.maxk, g.z, sink.Lk);
#pragma unroll
    for (int t = 0; t < TT; ++t) sink.acc[t] = 0.0;
    sph_point<LCAP, KCAP>(M, g, sink);
    const double E = exp(-0.5 * g.z);
    if (q < Q) {
#pragma unroll
        for (int t = 0; t < TT; ++t)
            if (t < tcount) out[(int64_t)t * Q + q] = in ? E * sink.acc[t] : __builtin_nan("");
    }
}

// ---------------------------------------------------------------------------------------------
// K2 fast path: exact template orders (L = MAXL, K = MAXK), one degree group.  Differences to the generic
// kernel above, all aimed at keeping the fp64 VALU busy (the kernel is VALU-bound):
//  * the recurrence table c[j][m] and the coefficient tile are staged in LDS once per workgroup and read as
//    wave-uniform (broadcast) ds_read_b128 - no dependent scalar loads in the inner loop;
//  * the triangular start-up (degrees j <= L) is unrolled at compile time, so the main degree loop has no
//    branches: 2 fp64 ops per (degree, order);
//  * the loop over l is unrolled, so each contraction knows l at compile time (no predication).
template <int L, int K, int TT>
struct FastEval {
    const double* shC;      // LDS [TT][L*L*K] coefficient tile, [t][r = l(l+1)+m][k]
    double Lk[K];
    double acc[TT];
    template <int l, typename CT>
    __device__ __forceinline__ void consume(const CT* cur, const double* cm, const double* sm)
    {
        constexpr int r0 = l * (l + 1);
        constexpr int NB = L * L * K;
#pragma unroll
        for (int m = 0; m <= l; ++m) {
            const double pc = (double)cur[m] * cm[m];
            const double ps = (double)cur[m] * sm[m];
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const double* cp = shC + t * NB + (r0 + m) * K;
                const double* cn = shC + t * NB + (r0 - m) * K;
                double Sp = 0.0, Sm = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    Sp = fma(cp[k], Lk[k], Sp);
                    if (m > 0) Sm = fma(cn[k], Lk[k], Sm);
                }
                acc[t] = fma(pc, Sp, acc[t]);
                if (m > 0) acc[t] = fma(ps, Sm, acc[t]);
            }
        }
    }
};

template <int L, int K, int TT, int l>
struct ConsumeAt {
    // consume degree l if its integer part equals j (start-up phase only: j <= L)
    template <typename CT>
    __device__ static __forceinline__ void run(FastEval<L, K, TT>& E, const int* nvl, int j, const CT* cur,
                                               const double* cm, const double* sm)
    {
        if (nvl[l] == j) E.template consume<l>(cur, cm, sm);
        if constexpr (l + 1 < L) ConsumeAt<L, K, TT, l + 1>::run(E, nvl, j, cur, cm, sm);
    }
};

template <int L, int K, int TT, int l>
struct MainSegments {
    template <typename CT>
    __device__ static __forceinline__ void run(FastEval<L, K, TT>& E, const CT* shc, const int* nvl, int& j, CT x,
                                               CT* cur, CT* prev, const double* cm, const double* sm)
    {
        const int jend = nvl[l];
        if (jend > L) {
#pragma unroll 2
            for (; j <= jend; ++j) {
                const CT* cj = shc + j * L;
#pragma unroll
                for (int m = 0; m < L; ++m) {
                    const CT nw = fma(x, cur[m], -(cj[m] * prev[m]));
                    prev[m] = cur[m];
                    cur[m] = nw;
                }
            }
            E.template consume<l>(cur, cm, sm);
        }
        if constexpr (l + 1 < L) MainSegments<L, K, TT, l + 1>::run(E, shc, nvl, j, x, cur, prev, cm, sm);
    }
};

// CT = arithmetic type of the Legendre chains: double, or float for the fp32 variant of BASELINE configs[4]'s tolerance
// sweep (seeds, trigonometric and Laguerre factors and the contraction with the coefficients stay fp64)
template <int L, int K, int TT, typename CT>
__global__ __launch_bounds__(BLOCK) void k_eval_sph_fast(SphDev M, int64_t Q, const double* __restrict__ lat,
                                                         const double* __restrict__ lon, const double* __restrict__ alt,
                                                         int tcount, const double* __restrict__ Cp,
                                                         const unsigned char* __restrict__ mask, int F, double tol,
                                                         double* __restrict__ out)
{
    extern __shared__ __align__(16) double sh[];
    constexpr int NB = L * L * K;
    const SphGroupDev G = M.groups[0];
    const int nj = G.nvmax + 1;
    CT* shc = reinterpret_cast<CT*>(sh);                // [nj][L] recurrence table in the chain's arithmetic type
    double* shC = sh + ((nj * L + 1) & ~1);             // [TT][NB]
    int* nvl = reinterpret_cast<int*>(shC + TT * NB);   // [L]
    for (int i = threadIdx.x; i < nj * L; i += BLOCK) shc[i] = (CT)G.c[i];
    for (int i = threadIdx.x; i < TT * NB; i += BLOCK) shC[i] = i < tcount * NB ? Cp[i] : 0.0;
    for (int j = threadIdx.x; j < nj; j += BLOCK) {
        const int l = G.pick[j];
        if (l >= 0) nvl[l] = j;
    }
    __syncthreads();

    const int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t qc = q < Q ? q : Q - 1;
    const Geom g = sph_geom(M, lat[qc], lon[qc], alt[qc]);
    bool in = true;
    if (F > 0) {
        in = mask[qc] != 0;
        if (!__any(in && q < Q)) {
            if (q < Q)
                for (int t = 0; t < tcount; ++t) out[(int64_t)t * Q + q] = __builtin_nan("");
            return;
        }
    }
    FastEval<L, K, TT> E;
    E.shC = shC;
    laguerre<K>(K, g.z, E.Lk);
#pragma unroll
    for (int t = 0; t < TT; ++t) E.acc[t] = 0.0;
    double cm[L], sm[L];
    cm[0] = 1.0;
    sm[0] = 0.0;
#pragma unroll
    for (int m = 1; m < L; ++m) {
        cm[m] = cm[m - 1] * g.cphi - sm[m - 1] * g.sphi;
        sm[m] = sm[m - 1] * g.cphi + cm[m - 1] * g.sphi;
    }
    const CT x = (CT)g.x;
    const double zz = 0.5 * (1.0 - g.x);
    const bool intseed = (G.nterms == 0);
    CT cur[L], prev[L];
#pragma unroll
    for (int m = 0; m < L; ++m) { cur[m] = (CT)0.0; prev[m] = (CT)0.0; }
    double pmm = 1.0, spow = 1.0;
    // ---- start-up: degrees j = 0 .. L, compile-time triangular structure ------------------------------
#pragma unroll
    for (int j = 0; j <= L; ++j) {
#pragma unroll
        for (int m = 0; m < L; ++m) {
            if (j > m + 1) {
                if (j < nj) {
                    const CT nw = fma(x, cur[m], -(shc[j * L + m] * prev[m]));
                    prev[m] = cur[m];
                    cur[m] = nw;
                }
            } else if (j == m) {
                if (m > 0) { pmm *= -(2.0 * m - 1.0) * g.s; spow *= g.s; }
                if (intseed) cur[m] = (CT)pmm;
                else cur[m] = (CT)(G.pref[m] * spow * hyp_series(G.q + (size_t)m * G.nterms, G.nterms, zz));
            } else if (j == m + 1) {
                prev[m] = cur[m];
                if (intseed) cur[m] = (CT)(g.x * (2.0 * m + 1.0) * pmm);
                else cur[m] = (CT)(G.pref[L + m] * spow * hyp_series(G.q + (size_t)(L + m) * G.nterms, G.nterms, zz));
            }
        }
        ConsumeAt<L, K, TT, 0>::run(E, nvl, j, cur, cm, sm);
    }
    // ---- main: all chains in recurrence mode; one segment per degree l --------------------------------
    int j = L + 1;
    MainSegments<L, K, TT, 0>::run(E, shc, nvl, j, x, cur, prev, cm, sm);
    const double Ez = exp(-0.5 * g.z);
    if (q < Q) {
#pragma unroll
        for (int t = 0; t < TT; ++t)
            if (t < tcount) out[(int64_t)t * Q + q] = in ? Ez * E.acc[t] : __builtin_nan("");
    }
}

// Cp[t][r*maxk + k] = C[t][k*L2 + r] * scale[r]
__device__ __forceinline__ void prep_coef_body(int T, int maxk, int L2, const double* __restrict__ C,
                                               const double* __restrict__ scale, double* __restrict__ Cp, int64_t bid)
{
    const int N = maxk * L2;
    const int64_t i = bid * blockDim.x + threadIdx.x;
    if (i >= (int64_t)T * N) return;
    const int t = (int)(i / N), n = (int)(i % N);
    const int r = n / maxk, k = n % maxk;
    Cp[i] = C[(int64_t)t * N + k * L2 + r] * scale[r];
}

__global__ void k_prep_coef(int T, int maxk, int L2, const double* __restrict__ C, const double* __restrict__ scale,
                            double* __restrict__ Cp)
{
    prep_coef_body(T, maxk, L2, C, scale, Cp, (int64_t)blockIdx.x);
}

// the two preparations of a masked evaluation call in one launch: blocks 0-3 the hull buffer, the others the coefficients
__global__ __launch_bounds__(256) void k_prep_hull_coef(int F, const double* __restrict__ eq, double* __restrict__ hullbuf, int T,
                                                        int maxk, int L2, const double* __restrict__ C,
                                                        const double* __restrict__ scale, double* __restrict__ Cp)
{
    if (blockIdx.x < HULL_PREP_BLOCKS) prep_hull_body(F, eq, hullbuf, (int)blockIdx.x, HULL_PREP_BLOCKS);
    else prep_coef_body(T, maxk, L2, C, scale, Cp, (int64_t)blockIdx.x - HULL_PREP_BLOCKS);
}

__global__ void k_transform_sph(SphDev M, int64_t P, const double* __restrict__ lat, const double* __restrict__ lon,
                                const double* __restrict__ alt, double* z, double* th, double* ph)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const Geom g = sph_geom(M, lat[p], lon[p], alt[p]);
    z[p] = g.z;
    th[p] = acos(g.x);
    ph[p] = atan2(g.Ry, g.Rx);
}

// ---------------------------------------------------------------------------------------------
// Gaussian RBF model (radbasfun.py:83-112): A[p,n] = exp(-|R_p - c_n|^2 / eps^2), R in ECEF metres
__global__ __launch_bounds__(BLOCK) void k_basis_rbf(RbfDev M, int64_t P, const double* __restrict__ lat,
                                                     const double* __restrict__ lon, const double* __restrict__ alt,
                                                     double* __restrict__ A, int64_t ld_p, int64_t ld_n)
{
    const int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= P) return;
    double X, Y, Z;
    geodetic2ecef(lat[p], lon[p], alt[p], X, Y, Z);
    const double* __restrict__ c = M.centers;
    for (int n = 0; n < M.N; ++n) {
        const double dx = X - c[3 * n], dy = Y - c[3 * n + 1], dz = Z - c[3 * n + 2];
        const double r2 = dx * dx + dy * dy + dz * dz;
        A[p * ld_p + (int64_t)n * ld_n] = exp(-r2 * M.inv_eps2);
    }
}

template <int TT>
__global__ __launch_bounds__(BLOCK) void k_eval_rbf(RbfDev M, int64_t Q, const double* __restrict__ lat,
                                                    const double* __restrict__ lon, const double* __restrict__ alt,
                                                    int tcount, const double* __restrict__ C,
                                                    const unsigned char* __restrict__ mask, int F, double tol,
                                                    double* __restrict__ out)
{
    const int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t qc = q < Q ? q : Q - 1;
    double X, Y, Z;
    geodetic2ecef(lat[qc], lon[qc], alt[qc], X, Y, Z);
    bool in = true;
    if (F > 0) {
        in = mask[qc] != 0;
        if (!__any(in && q < Q)) {
            if (q < Q)
                for (int t = 0; t < tcount; ++t) out[(int64_t)t * Q + q] = __builtin_nan("");
            return;
        }
    }
    double acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[t] = 0.0;
    const double* __restrict__ c = M.centers;
    for (int n = 0; n < M.N; ++n) {
        const double dx = X - c[3 * n], dy = Y - c[3 * n + 1], dz = Z - c[3 * n + 2];
        const double e = exp(-(dx * dx + dy * dy + dz * dz) * M.inv_eps2);
#pragma unroll
        for (int t = 0; t < TT; ++t)
            if (t < tcount) acc[t] = fma(e, C[(size_t)t * M.N + n], acc[t]);
    }
    if (q < Q) {
#pragma unroll
        for (int t = 0; t < TT; ++t)
            if (t < tcount) out[(int64_t)t * Q + q] = in ? acc[t] : __builtin_nan("");
    }
}

__global__ void k_transform_rbf(int64_t P, const double* __restrict__ lat, const double* __restrict__ lon,
                                const double* __restrict__ alt, double* X, double* Y, double* Z)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    geodetic2ecef(lat[p], lon[p], alt[p], X[p], Y[p], Z[p]);
}

inline unsigned nblocks(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

// (LCAP, KCAP) instantiations: default order; C5 order; larger orders
template <int LCAP, int KCAP>
int launch_basis_sph(vi_model* m, int64_t P, const double* lat, const double* lon, const double* alt, double* A,
                     int64_t ld_p, int64_t ld_n)
{
    hipLaunchKernelGGL((k_basis_sph<LCAP, KCAP>), dim3(nblocks(P, BLOCK)), dim3(BLOCK), 0, m->ctx->stream, m->sph, P,
                       lat, lon, alt, A, ld_p, ld_n);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

template <int LCAP, int KCAP>
int launch_eval_sph(vi_model* m, int64_t Q, const double* lat, const double* lon, const double* alt, int64_t T,
                    const double* Cp, const unsigned char* hull, int F, double tol, double* out)
{
    const int N = m->N;
    int64_t t = 0;
    while (t < T) {
        const int64_t left = T - t;
        if (left >= 4) {
            hipLaunchKernelGGL((k_eval_sph<LCAP, KCAP, 4>), dim3(nblocks(Q, BLOCK)), dim3(BLOCK), 0, m->ctx->stream,
                               m->sph, Q, lat, lon, alt, 4, Cp + t * N, hull, F, tol, out + t * Q);
            t += 4;
        } else {
            hipLaunchKernelGGL((k_eval_sph<LCAP, KCAP, 1>), dim3(nblocks(Q, BLOCK)), dim3(BLOCK), 0, m->ctx->stream,
                               m->sph, Q, lat, lon, alt, 1, Cp + t * N, hull, F, tol, out + t * Q);
            t += 1;
        }
        VI_HIP(hipGetLastError());
    }
    return VI_OK;
}


template <int L, int K, typename CT>
int launch_eval_sph_fast(vi_model* m, int64_t Q, const double* lat, const double* lon, const double* alt, int64_t T,
                         const double* Cp, const unsigned char* hull, int F, double tol, double* out)
{
    const int N = m->N;
    const int nj = m->nvmax0 + 1;
    auto shm = [&](int TT) { return (size_t)(((nj * L + 1) & ~1) + TT * N) * sizeof(double) + L * sizeof(int) + 16; };
    // per call, not cached: the attribute is per device and several device contexts may live in one process
    VI_HIP(hipFuncSetAttribute((const void*)k_eval_sph_fast<L, K, 16, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VI_HIP(hipFuncSetAttribute((const void*)k_eval_sph_fast<L, K, 4, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VI_HIP(hipFuncSetAttribute((const void*)k_eval_sph_fast<L, K, 1, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    // timestep tiles of 16 / 4 / 1: the basis is recomputed once per tile, so a wide tile amortises it and the
    // contraction (2N flop per point-timestep) dominates.  Measured at 128^3, N = 144: 1.4e10 / 3.9e10 / 6.2e10
    // point-timesteps/s for tiles of 1 / 4 / 16 (a tile of 8 is slower than 16).  Whole tiles of 16 timesteps normally
    // never get here: vi_eval_mfma.hip contracts them on the matrix cores (1.2e11).
    const bool wide_ok = shm(16) <= 60 * 1024;
    int64_t t = 0;
    while (t < T) {
        if (wide_ok && T - t >= 16) {
            hipLaunchKernelGGL((k_eval_sph_fast<L, K, 16, CT>), dim3(nblocks(Q, BLOCK)), dim3(BLOCK), shm(16), m->ctx->stream,
                               m->sph, Q, lat, lon, alt, 16, Cp + t * N, hull, F, tol, out + t * Q);
            t += 16;
        } else if (T - t >= 4) {
            hipLaunchKernelGGL((k_eval_sph_fast<L, K, 4, CT>), dim3(nblocks(Q, BLOCK)), dim3(BLOCK), shm(4), m->ctx->stream,
                               m->sph, Q, lat, lon, alt, 4, Cp + t * N, hull, F, tol, out + t * Q);
            t += 4;
        } else {
            hipLaunchKernelGGL((k_eval_sph_fast<L, K, 1, CT>), dim3(nblocks(Q, BLOCK)), dim3(BLOCK), shm(1), m->ctx->stream,
                               m->sph, Q, lat, lon, alt, 1, Cp + t * N, hull, F, tol, out + t * Q);
            t += 1;
        }
        VI_HIP(hipGetLastError());
    }
    return VI_OK;
}

// records HIP events around the evaluation kernel launches of one vi_eval_f64 call (see vi_eval_kernel_ms)
struct EvalTimer {
    vi_ctx* c;
    explicit EvalTimer(vi_ctx* ctx) : c(ctx)
    {
        c->evk_valid = c->evk_enabled && hipEventRecord(c->evk0, c->stream) == hipSuccess;
    }
    ~EvalTimer()
    {
        if (c->evk_valid) c->evk_valid = hipEventRecord(c->evk1, c->stream) == hipSuccess;
    }
};

}  // namespace
int vi_eval_sph_mfma(vi_model* m, int64_t Q, const double* lat, const double* lon, const double* alt, int64_t T,
                     const double* Cp, const unsigned char* hull, int F, double* out, int64_t* done);
int vi_eval_sph_split(vi_model* m, int64_t Q, const double* lat, const double* lon, const double* alt, int64_t T,
                      const double* Cp, const unsigned char* hull, int F, double* out, int* handled);
int vi_eval_resident_mfma(vi_ctx* c, int N, int64_t Q, int64_t T, const double* d_Y, const double* d_C, double* d_out, int* handled);
namespace {

bool use_fast_eval()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VINTERP_EVAL");
        v = (e && !strcmp(e, "generic")) ? 0 : 1;
    }
    return v == 1;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" int vi_basis_f64(vi_model* m, int64_t P, const double* d_lat, const double* d_lon, const double* d_alt,
                            double* d_A, int64_t ld_p, int64_t ld_n)
{
    VI_REQUIRE(m && d_lat && d_lon && d_alt && d_A, "null argument");
    VI_REQUIRE(P >= 0, "negative point count");
    if (P == 0) return VI_OK;
    VI_HIP(hipSetDevice(m->ctx->device));
    if (m->kind == VI_MODEL_SPHHARMLAG) {
        const int L = m->sph.maxl, K = m->sph.maxk;
        if (L <= 6 && K <= 4) return launch_basis_sph<6, 4>(m, P, d_lat, d_lon, d_alt, d_A, ld_p, ld_n);
        if (L <= 12 && K <= 8) return launch_basis_sph<12, 8>(m, P, d_lat, d_lon, d_alt, d_A, ld_p, ld_n);
        if (L <= 24 && K <= 16) return launch_basis_sph<24, 16>(m, P, d_lat, d_lon, d_alt, d_A, ld_p, ld_n);
        vi_set_error("vi_basis_f64: order MAXL=%d MAXK=%d beyond the compiled limits (24, 16)", L, K);
        return VI_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_basis_rbf, dim3(nblocks(P, BLOCK)), dim3(BLOCK), 0, m->ctx->stream, m->rbf, P, d_lat, d_lon,
                       d_alt, d_A, ld_p, ld_n);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

extern "C" int vi_grad_basis_f64(vi_model* m, int64_t P, const double* d_lat, const double* d_lon, const double* d_alt,
                                 double* d_G, int64_t ld_p, int64_t ld_c, int64_t ld_n)
{
    VI_REQUIRE(m && d_lat && d_lon && d_alt && d_G, "null argument");
    VI_REQUIRE(P >= 0, "negative point count");
    if (m->kind != VI_MODEL_SPHHARMLAG || !m->sph.scale1 || !m->sph.nu) {
        vi_set_error("vi_grad_basis_f64: only the sphharmlag model provides a gradient basis");
        return VI_ERR_UNSUPPORTED;
    }
    if (P == 0) return VI_OK;
    VI_HIP(hipSetDevice(m->ctx->device));
    const int L = m->sph.maxl, K = m->sph.maxk;
    const dim3 grid(nblocks(P, BLOCK)), block(BLOCK);
    if (L <= 6 && K <= 4)
        hipLaunchKernelGGL((k_grad_sph<6, 4, false>), grid, block, 0, m->ctx->stream, m->sph, P, d_lat, d_lon, d_alt, nullptr,
                           d_G, ld_p, ld_c, ld_n);
    else if (L <= 12 && K <= 8)
        hipLaunchKernelGGL((k_grad_sph<12, 8, false>), grid, block, 0, m->ctx->stream, m->sph, P, d_lat, d_lon, d_alt, nullptr,
                           d_G, ld_p, ld_c, ld_n);
    else {
        vi_set_error("vi_grad_basis_f64: order MAXL=%d MAXK=%d beyond the compiled limits (12, 8)", L, K);
        return VI_ERR_UNSUPPORTED;
    }
    VI_HIP(hipGetLastError());
    return VI_OK;
}

// Gradient of the fitted parameter: out[q][c] = sum_n grad_basis[q][c][n] * C[n], c = z, theta, phi components
// (sphharmlag.py:148-184 contracted with one coefficient vector; the (Q, 3, N) array is never formed).
extern "C" int vi_eval_grad_f64(vi_model* m, int64_t Q, const double* d_lat, const double* d_lon, const double* d_alt,
                                const double* d_C, double* d_out)
{
    VI_REQUIRE(m && d_lat && d_lon && d_alt && d_C && d_out, "null argument");
    VI_REQUIRE(Q >= 0, "negative point count");
    if (m->kind != VI_MODEL_SPHHARMLAG || !m->sph.scale1 || !m->sph.nu) {
        vi_set_error("vi_eval_grad_f64: only the sphharmlag model provides a gradient basis");
        return VI_ERR_UNSUPPORTED;
    }
    if (Q == 0) return VI_OK;
    VI_HIP(hipSetDevice(m->ctx->device));
    const int L = m->sph.maxl, K = m->sph.maxk;
    const dim3 grid(nblocks(Q, BLOCK)), block(BLOCK);
    if (L <= 6 && K <= 4)
        hipLaunchKernelGGL((k_grad_sph<6, 4, true>), grid, block, 0, m->ctx->stream, m->sph, Q, d_lat, d_lon, d_alt, d_C, d_out,
                           (int64_t)3, (int64_t)1, (int64_t)0);
    else if (L <= 12 && K <= 8)
        hipLaunchKernelGGL((k_grad_sph<12, 8, true>), grid, block, 0, m->ctx->stream, m->sph, Q, d_lat, d_lon, d_alt, d_C, d_out,
                           (int64_t)3, (int64_t)1, (int64_t)0);
    else {
        vi_set_error("vi_eval_grad_f64: order MAXL=%d MAXK=%d beyond the compiled limits (12, 8)", L, K);
        return VI_ERR_UNSUPPORTED;
    }
    VI_HIP(hipGetLastError());
    return VI_OK;
}

// Standard error of the fitted parameter from the coefficient covariance: err[q] = sqrt(a_q^T dC a_q), a_q = basis row
// of point q (first-order error propagation; the `calcerr` output the reference's Estimate.__call__ advertises but never
// computes, estimate.py:139-145).  Chunks of points: basis tile A (K1) -> B = A dC (rocBLAS fp64 GEMM, MFMA) ->
// row-wise dot.  2 N^2 flop per point: MFMA-bound.
namespace {
__global__ void k_rowdot_sqrt(int64_t P, int N, const double* __restrict__ A, const double* __restrict__ B,
                              double* __restrict__ out)
{
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= P) return;
    double acc = 0.0;
    for (int n = lane; n < N; n += 64) acc = fma(A[p * N + n], B[p * N + n], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane == 0) out[p] = sqrt(acc);          // a negative variance (indefinite dC) gives NaN, as np.sqrt would
}
}  // namespace

extern "C" int vi_eval_err_f64(vi_model* m, int64_t Q, const double* d_lat, const double* d_lon, const double* d_alt,
                               const double* d_dC, double* d_out)
{
    VI_REQUIRE(m && d_lat && d_lon && d_alt && d_dC && d_out, "null argument");
    VI_REQUIRE(Q >= 0, "negative point count");
    if (Q == 0) return VI_OK;
    VI_HIP(hipSetDevice(m->ctx->device));
    const int N = m->N;
    const int64_t chunk = 1 << 16;
    void* ws = nullptr;
    int rc = vi_ctx_workspace(m->ctx, (size_t)2 * chunk * N * sizeof(double), &ws);
    if (rc != VI_OK) return rc;
    double* A = (double*)ws;
    double* B = A + chunk * N;
    const double one = 1.0, zero = 0.0;
    for (int64_t q0 = 0; q0 < Q; q0 += chunk) {
        const int64_t qc = (Q - q0) < chunk ? (Q - q0) : chunk;
        rc = vi_basis_f64(m, qc, d_lat + q0, d_lon + q0, d_alt + q0, A, N, 1);            // row-major (qc, N)
        if (rc != VI_OK) return rc;
        // row-major B (qc x N) = A (qc x N) dC (N x N)  <=>  column-major B^T (N x qc) = dC^T (N x N) A^T (N x qc)
        VI_ROCBLAS(rocblas_dgemm(m->ctx->blas, rocblas_operation_transpose, rocblas_operation_none, N, (rocblas_int)qc, N, &one,
                                 d_dC, N, A, N, &zero, B, N));
        hipLaunchKernelGGL(k_rowdot_sqrt, dim3(nblocks(qc, 4)), dim3(256), 0, m->ctx->stream, qc, N, A, B, d_out + q0);
        VI_HIP(hipGetLastError());
    }
    return VI_OK;
}

// Arithmetic of the Legendre degree recurrences of the fused evaluation: 0 = fp64 (default), 1 = fp32 chains (seeds,
// trigonometric and Laguerre factors and the contraction stay fp64).  BASELINE configs[4]'s fp32-vs-fp64 tolerance sweep.
extern "C" int vi_model_set_eval_precision(vi_model* m, int32_t chain_f32)
{
    VI_REQUIRE(m, "null model");
    VI_REQUIRE(chain_f32 == 0 || chain_f32 == 1, "precision must be 0 (fp64) or 1 (fp32 chains)");
    if (chain_f32 && m->kind != VI_MODEL_SPHHARMLAG) {
        vi_set_error("vi_model_set_eval_precision: the fp32 variant exists for the sphharmlag model only");
        return VI_ERR_UNSUPPORTED;
    }
    m->chain_f32 = chain_f32 != 0;
    return VI_OK;
}

extern "C" int vi_transform_f64(vi_model* m, int64_t P, const double* d_lat, const double* d_lon, const double* d_alt,
                                double* d_c0, double* d_c1, double* d_c2)
{
    VI_REQUIRE(m && d_lat && d_lon && d_alt && d_c0 && d_c1 && d_c2, "null argument");
    if (P <= 0) return P == 0 ? VI_OK : VI_ERR_INVALID;
    VI_HIP(hipSetDevice(m->ctx->device));
    if (m->kind == VI_MODEL_SPHHARMLAG)
        hipLaunchKernelGGL(k_transform_sph, dim3(nblocks(P, BLOCK)), dim3(BLOCK), 0, m->ctx->stream, m->sph, P, d_lat,
                           d_lon, d_alt, d_c0, d_c1, d_c2);
    else
        hipLaunchKernelGGL(k_transform_rbf, dim3(nblocks(P, BLOCK)), dim3(BLOCK), 0, m->ctx->stream, P, d_lat, d_lon,
                           d_alt, d_c0, d_c1, d_c2);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

// ---- evaluation of many timesteps on one grid from a RESIDENT basis (BASELINE configs[3]: a GPU's share of 10 000 timesteps on
// one 256^3 grid).  vi_eval_f64 recomputes the basis of a point for every tile of 32 timesteps - right for one timestep or a
// few, and it keeps the matrix cores waiting for the recurrences half of the time.  With hundreds of
// timesteps on the same grid the basis matrix is worth keeping: N x Q doubles - 19 GB at the default order on 256^3, a
// fifteenth of this GPU's HBM - assembled once by K1 (vi_basis_f64, basis-major so that every access is a contiguous run of
// points), the rows of points outside the hull set to NaN, so that the mask costs nothing afterwards: NaN times anything is
// NaN.  The evaluation is then a plain product out(Q x T) = Y(Q x N) C(N x T) - the library's GEMM, in tiles of 128 timesteps
// (tools/microbench/dgemm_eval_shape.hip: 59 TFLOP/s at 128, 32 at 64, 56 at 256; Y is read once per tile: 0.15 B/flop).
namespace {
__global__ void k_mask_basis(int64_t Q, int N, const unsigned char* __restrict__ mask, double* __restrict__ Y)
{
    const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (q >= Q || mask[q] != 0) return;
    const double nan = __builtin_nan("");
    for (int n = 0; n < N; ++n) Y[(int64_t)n * Q + q] = nan;
}
}  // namespace

namespace {
// the hull pass of a call: on the matrix cores (k_hull_mask_mx) unless VINTERP_HULL=fp32 asks for the packed-fp32 loop
void launch_hull_mask(vi_model* m, int64_t Q, const double* d_lat, const double* d_lon, const double* d_alt, int F, double tol)
{
    static const bool fp32 = [] {
        const char* e = getenv("VINTERP_HULL");
        return e && strcmp(e, "fp32") == 0;
    }();
    if (fp32)
        hipLaunchKernelGGL(k_hull_mask, dim3(nblocks(Q, BLOCK * HULL_PP)), dim3(BLOCK), 0, m->ctx->stream, Q, d_lat, d_lon, d_alt,
                           m->d_hull, F, tol, m->d_mask);
    else
        hipLaunchKernelGGL(k_hull_mask_mx<HULL_MX_PP>, dim3(nblocks(Q, BLOCK * HULL_MX_PP)), dim3(BLOCK), 0, m->ctx->stream, Q,
                           d_lat, d_lon, d_alt, m->d_hull, F, tol, m->d_mask);
}
}  // namespace

extern "C" int vi_eval_basis_f64(vi_model* m, int64_t Q, const double* d_lat, const double* d_lon, const double* d_alt,
                                 const double* d_hull_eq, int32_t F, double hull_tol, double* d_Y)
{
    VI_REQUIRE(m && d_lat && d_lon && d_alt && d_Y, "null argument");
    VI_REQUIRE(Q >= 0 && F >= 0, "negative size");
    VI_REQUIRE(F == 0 || d_hull_eq, "hull facet count given without facet equations");
    if (Q == 0) return VI_OK;
    int rc = vi_basis_f64(m, Q, d_lat, d_lon, d_alt, d_Y, 1, Q);
    if (rc != VI_OK || F == 0) return rc;
    const size_t need = hull_buf_bytes((size_t)F);
    if (need > m->hull_bytes) {
        VI_HIP(hipStreamSynchronize(m->ctx->stream));
        if (m->d_hull) VI_HIP(hipFree(m->d_hull));
        m->d_hull = nullptr;
        m->hull_bytes = 0;
        VI_HIP(hipMalloc((void**)&m->d_hull, need));
        m->hull_bytes = need;
    }
    hipLaunchKernelGGL(k_prep_hull, dim3(HULL_PREP_BLOCKS), dim3(256), 0, m->ctx->stream, (int)F, d_hull_eq, m->d_hull);
    VI_HIP(hipGetLastError());
    if ((size_t)Q > m->mask_bytes) {
        VI_HIP(hipStreamSynchronize(m->ctx->stream));
        if (m->d_mask) VI_HIP(hipFree(m->d_mask));
        m->d_mask = nullptr;
        m->mask_bytes = 0;
        VI_HIP(hipMalloc((void**)&m->d_mask, (size_t)Q));
        m->mask_bytes = (size_t)Q;
    }
    launch_hull_mask(m, Q, d_lat, d_lon, d_alt, (int)F, hull_tol);
    hipLaunchKernelGGL(k_mask_basis, dim3(nblocks(Q, 256)), dim3(256), 0, m->ctx->stream, Q, m->N, m->d_mask, d_Y);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

extern "C" int vi_eval_resident_f64(vi_model* m, int64_t Q, int64_t T, const double* d_Y, const double* d_C, double* d_out)
{
    VI_REQUIRE(m && d_Y && d_C && d_out, "null argument");
    VI_REQUIRE(Q >= 0 && T >= 0, "negative size");
    if (Q == 0 || T == 0) return VI_OK;
    vi_ctx* c = m->ctx;
    VI_HIP(hipSetDevice(c->device));
    const int N = m->N;
    {
        EvalTimer timer(c);
        int handled = 0;
        const int rc = vi_eval_resident_mfma(c, N, Q, T, d_Y, d_C, d_out, &handled);      // K2r (vi_eval_resident.hip)
        if (rc != VI_OK || handled) return rc;
    }
    // shapes outside K2r's (Q not a multiple of 4, orders whose coefficient tile exceeds the LDS): the library's product
    const double one = 1.0, zero = 0.0;
    const int64_t TT = 128;                          // timesteps per product
    const int64_t QQ = (int64_t)1 << 30;             // points per product (the library's dimensions are 32-bit)
    EvalTimer timer(c);
    for (int64_t t0 = 0; t0 < T; t0 += TT) {
        const int64_t tc = (T - t0) < TT ? (T - t0) : TT;
        for (int64_t q0 = 0; q0 < Q; q0 += QQ) {
            const int64_t qc = (Q - q0) < QQ ? (Q - q0) : QQ;
            // column-major: out(qc x tc, ld Q) = Y(qc x N, ld Q) * C(N x tc, ld N)
            VI_ROCBLAS(rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_none, (rocblas_int)qc, (rocblas_int)tc, N,
                                     &one, d_Y + q0, (rocblas_int)Q, d_C + t0 * N, N, &zero, d_out + t0 * Q + q0, (rocblas_int)Q));
        }
    }
    return VI_OK;
}

extern "C" int vi_eval_f64(vi_model* m, int64_t Q, const double* d_lat, const double* d_lon, const double* d_alt,
                           int64_t T, const double* d_C, const double* d_hull_eq, int32_t F, double hull_tol,
                           double* d_out)
{
    VI_REQUIRE(m && d_lat && d_lon && d_alt && d_C && d_out, "null argument");
    VI_REQUIRE(Q >= 0 && T >= 0 && F >= 0, "negative size");
    VI_REQUIRE(F == 0 || d_hull_eq, "hull facet count given without facet equations");
    if (Q == 0 || T == 0) return VI_OK;
    VI_HIP(hipSetDevice(m->ctx->device));
    const int N = m->N;
    const bool sph = m->kind == VI_MODEL_SPHHARMLAG;
    if (sph) {
        const size_t need = (size_t)T * N * sizeof(double);
        if (need > m->coef_bytes) {
            if (m->d_coef) VI_HIP(hipFree(m->d_coef));
            m->d_coef = nullptr;
            m->coef_bytes = 0;
            VI_HIP(hipMalloc((void**)&m->d_coef, need));
            m->coef_bytes = need;
        }
    }
    const int L2 = sph ? m->sph.maxl * m->sph.maxl : 0;
    if (F > 0) {
        const size_t need = hull_buf_bytes((size_t)F);
        if (need > m->hull_bytes) {
            VI_HIP(hipStreamSynchronize(m->ctx->stream));
            if (m->d_hull) VI_HIP(hipFree(m->d_hull));
            m->d_hull = nullptr;
            m->hull_bytes = 0;
            VI_HIP(hipMalloc((void**)&m->d_hull, need));
            m->hull_bytes = need;
        }
        if (sph)            // the hull buffer and the coefficients of the call in one launch
            hipLaunchKernelGGL(k_prep_hull_coef, dim3(HULL_PREP_BLOCKS + nblocks((int64_t)T * N, 256)), dim3(256), 0, m->ctx->stream,
                               (int)F, d_hull_eq, m->d_hull, (int)T, m->sph.maxk, L2, d_C, m->sph.scale, m->d_coef);
        else
            hipLaunchKernelGGL(k_prep_hull, dim3(HULL_PREP_BLOCKS), dim3(256), 0, m->ctx->stream, (int)F, d_hull_eq, m->d_hull);
        VI_HIP(hipGetLastError());
        if ((size_t)Q > m->mask_bytes) {
            VI_HIP(hipStreamSynchronize(m->ctx->stream));
            if (m->d_mask) VI_HIP(hipFree(m->d_mask));
            m->d_mask = nullptr;
            m->mask_bytes = 0;
            VI_HIP(hipMalloc((void**)&m->d_mask, (size_t)Q));
            m->mask_bytes = (size_t)Q;
        }
        launch_hull_mask(m, Q, d_lat, d_lon, d_alt, (int)F, hull_tol);
        VI_HIP(hipGetLastError());
    }
    const unsigned char* d_mask = F > 0 ? m->d_mask : nullptr;
    if (sph) {
        if (F == 0) {
            hipLaunchKernelGGL(k_prep_coef, dim3(nblocks((int64_t)T * N, 256)), dim3(256), 0, m->ctx->stream, (int)T,
                               m->sph.maxk, L2, d_C, m->sph.scale, m->d_coef);
            VI_HIP(hipGetLastError());
        }
        const int L = m->sph.maxl, K = m->sph.maxk;
        EvalTimer timer(m->ctx);
        // whole tiles of 16 timesteps go to the matrix-core kernel (vi_eval_mfma.hip); the rest to the VALU kernels
        int64_t done = 0;
        if (use_fast_eval() && !m->chain_f32) {
            const int rc = vi_eval_sph_mfma(m, Q, d_lat, d_lon, d_alt, T, m->d_coef, d_mask, (int)F, d_out, &done);
            if (rc != VI_OK) return rc;
            if (done == T) return VI_OK;
        }
        const int64_t Tr = T - done;
        const double* coef = m->d_coef + done * N;
        double* outp = d_out + done * Q;
        if (use_fast_eval()) {                  // high orders: the chains in groups (vi_eval_split.hip)
            int handled = 0;
            const int rc = vi_eval_sph_split(m, Q, d_lat, d_lon, d_alt, Tr, coef, d_mask, (int)F, outp, &handled);
            if (rc != VI_OK || handled) return rc;
        }
        if (use_fast_eval() && m->sph.ngroups == 1 && (size_t)(m->nvmax0 + 1) * L * 8 + (size_t)4 * N * 8 < 60 * 1024) {
            if (m->chain_f32) {     // fp32 Legendre chains (vi_model_set_eval_precision): the orders of the tolerance sweep
                if (L == 6 && K == 4) return launch_eval_sph_fast<6, 4, float>(m, Q, d_lat, d_lon, d_alt, Tr, coef, d_mask, F, hull_tol, outp);
                if (L == 2 && K == 8) return launch_eval_sph_fast<2, 8, float>(m, Q, d_lat, d_lon, d_alt, Tr, coef, d_mask, F, hull_tol, outp);
                vi_set_error("vi_eval_f64: no fp32-chain kernel for MAXL=%d MAXK=%d", L, K);
                return VI_ERR_UNSUPPORTED;
            }
#define VI_FAST(LL, KK) \
    if (L == LL && K == KK) return launch_eval_sph_fast<LL, KK, double>(m, Q, d_lat, d_lon, d_alt, Tr, coef, d_mask, F, hull_tol, outp)
            VI_FAST(6, 4);
            VI_FAST(2, 8);
            VI_FAST(3, 4);
            VI_FAST(4, 3);
            VI_FAST(3, 2);
            VI_FAST(12, 2);
            VI_FAST(12, 8);
#undef VI_FAST
        }
        if (L <= 6 && K <= 4)
            return launch_eval_sph<6, 4>(m, Q, d_lat, d_lon, d_alt, Tr, coef, d_mask, F, hull_tol, outp);
        if (L <= 12 && K <= 8)
            return launch_eval_sph<12, 8>(m, Q, d_lat, d_lon, d_alt, Tr, coef, d_mask, F, hull_tol, outp);
        if (L <= 24 && K <= 16)
            return launch_eval_sph<24, 16>(m, Q, d_lat, d_lon, d_alt, Tr, coef, d_mask, F, hull_tol, outp);
        vi_set_error("vi_eval_f64: order MAXL=%d MAXK=%d beyond the compiled limits (24, 16)", L, K);
        return VI_ERR_UNSUPPORTED;
    }
    EvalTimer timer(m->ctx);
    int64_t t = 0;
    // the exponential of a (point, centre) pair is the cost (~20 of the ~28 fp64 operations per pair): it is computed once per
    // tile of timesteps, so the tile is as wide as the registers allow (16 accumulators; a tile of 4 recomputed every
    // exponential four times for 16 timesteps: 10.7 ms instead of 3.4 at 1000 centres x 128^3 points)
    while (t < T) {
        const int64_t left = T - t;
        const int tc = left >= 16 ? 16 : left >= 4 ? 4 : 1;
#define VI_RBF(TT)                                                                                                        \
    hipLaunchKernelGGL(k_eval_rbf<TT>, dim3(nblocks(Q, BLOCK)), dim3(BLOCK), 0, m->ctx->stream, m->rbf, Q, d_lat, d_lon, \
                       d_alt, TT, d_C + t * N, d_mask, F, hull_tol, d_out + t * Q)
        if (tc == 16) VI_RBF(16);
        else if (tc == 4) VI_RBF(4);
        else VI_RBF(1);
#undef VI_RBF
        VI_HIP(hipGetLastError());
        t += tc;
    }
    return VI_OK;
}
