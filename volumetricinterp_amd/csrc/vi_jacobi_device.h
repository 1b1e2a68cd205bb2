// Device side of K3, shared by the kernels that run the in-LDS Jacobi solve: k_jacobi_solve (vi_jacobi.hip: one launch,
// one system per workgroup) and k_brent_warm (vi_brent.hip: a workgroup runs the whole root-finder iteration of a record,
// solve after solve).  One source, one arithmetic: the library is compiled with -ffp-contract=on (Makefile), so that
// multiply-adds are fused by the rules of the language - statement by statement - and not by an optimiser whose
// decisions depend on what a function is inlined into.
#pragma once
#include "vi_common.h"

#ifndef VI_STAMP
#define VI_STAMP(k)
#endif
#ifndef VI_ROT_COUNT
#define VI_ROT_COUNT(sweep, r0, r1, r2, r3)
#define VI_ROT_SWEEP(sweep)
#endif

namespace {

// A pair whose three elements all lie below TINY_FRACTION x (rcond x max|diag|) is left alone: it sits inside the subspace
// the truncation is going to drop, where the entries are rounding noise that the relative criterion would chase for ever.
// The fraction must be small: k coupled entries of size d hide an eigenvalue of up to k d, and with the fraction at 1 the
// kernel lost eigenvalues at 1.6 x the cut (rank 107 where exact arithmetic keeps 108, BASELINE configs[1] geometry).
constexpr double TINY_FRACTION = 0.0625;

// Termination tolerance of the all-pairs test in units of the rotation criterion (eps).  1 = the classical rule.
#ifndef VI_CONV_FACTOR
#define VI_CONV_FACTOR 1.0
#endif

// index of element (r, c) of a symmetric 4x4 block among its 10 unique elements: the diagonal first
__device__ __host__ __forceinline__ int j10(int r, int c)
{
    if (r == c) return r;
    if (r > c) { const int t = r; r = c; c = t; }
    return r == 0 ? 3 + c : (r == 1 ? 5 + c : 9);          // (0,1) 4  (0,2) 5  (0,3) 6  (1,2) 7  (1,3) 8  (2,3) 9
}

// LDS address of element (i, j) of the slot-indexed symmetric matrix, M = number of matches.  The M(M-1)/2
// off-diagonal super-blocks S_ab (a < b, block index k = b(b-1)/2 + a, the order the threads enumerate them in) are
// stored as sixteen planes E[4r + c][k] for element (4a + r, 4b + c); the diagonal blocks as ten planes D[j10][a].
// Consecutive lanes own consecutive blocks, so the block reads of a round are linear in the lane (bank-conflict free).
__device__ __host__ __forceinline__ int tri4(int i, int j, int M)
{
    const int a = i >> 2, r = i & 3, b = j >> 2, c = j & 3;
    const int nsb = (M * (M - 1)) >> 1;
    if (a == b) return 16 * nsb + j10(r, c) * M + a;
    if (a < b) return (4 * r + c) * nsb + ((b * (b - 1)) >> 1) + a;
    return (4 * c + r) * nsb + ((a * (a - 1)) >> 1) + b;
}

// Brent & Luk permutation of 2m places after every round: place 0 is fixed, the others advance along the ring
// 1 -> 2 -> 4 -> ... -> 2m-2 -> 2m-1 -> 2m-3 -> ... -> 3 -> 1.  Here the places hold UNITS (2M of them).
__device__ __forceinline__ int ring_next(int s, int m)
{
    if (s == 0) return 0;
    if (s == 1) return 2;
    if (s & 1) return s - 2;
    return s == 2 * m - 2 ? 2 * m - 1 : s + 2;
}
// slot permutation: unit u = slots (2u, 2u+1) moves as a whole
__device__ __forceinline__ int slot_next(int s, int M) { return 2 * ring_next(s >> 1, M) + (s & 1); }

// 1 / sqrt(x) for x in a safe range (no denormals, no overflow: the systems are scaled to max|X| in [1, 2) and rotated
// elements exceed the absolute floor): hardware estimate v_rsq_f64 (measured 2^-24.2, tools/microbench/
// rsq_f64_precision.hip) + ONE third-order step y (1 + e/2 + 3 e^2/8), e = 1 - x y^2 (remaining error ~e^3/3 = 5e-23):
// 6 dependent operations instead of the ~25 of the library sqrt / rsqrt, which also handle scaling and special values
// (two Newton steps would be 9; one alone leaves 4e-15).  The rotation set-up is a chain of dependent fp64 operations
// in ONE wave while the others wait at the barrier: it was 2700 of the 5600 cycles of a round.
__device__ __forceinline__ double fast_rsqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y, e * fma(0.375, e, 0.5), y);
}

// plane rotation of the pair (p, q): returns true and (c, s), the new diagonal entries, when the pair is to be rotated.
// With d = a_qq - a_pp, r = sqrt(d^2 + 4 a_pq^2) and w = cos^2(theta) = (r + |d|) / (2 r) in [1/2, 1]:
//   c = sqrt(w) = w / sqrt(w),   s = sgn(d a_pq) |a_pq| / (r c),   t = tan(theta) = s / c = sgn(d a_pq) |a_pq| / (r w),
// i.e. the classical t = sgn(d) 2 a_pq / (|d| + r) written with two reciprocal square roots and no division; nothing
// cancels (w is formed by an addition of positive terms), and c^2 + s^2 = 1 holds to rounding.
__device__ __forceinline__ bool rot_params(double app, double aqq, double apq, double drop, double abs_floor,
                                           double& c, double& s, double& npp, double& nqq)
{
    // branch-free: the two rotations of an inner round are independent chains of ~25 dependent fp64 operations each;
    // computed under `if`s they ran one after the other (divergent branches do not interleave), as selects the compiler
    // overlaps them.  A pair that is not rotated may produce inf / NaN intermediates (x = 0); they are discarded.
    const double eps2 = 2.220446049250313e-16 * 2.220446049250313e-16;
    const double aa = fabs(apq);
    const bool tiny = fmax(fmax(fabs(app), fabs(aqq)), aa) < TINY_FRACTION * drop;
    const bool rot = aa * aa > eps2 * fabs(app * aqq) && aa > abs_floor && !tiny;
    const double d = aqq - app;
    const double ri = fast_rsqrt(fma(d, d, 4.0 * apq * apq));      // 1 / r
    const double w = fma(0.5 * fabs(d), ri, 0.5);
    const double rw = fast_rsqrt(w);
    const double q = copysign(aa * ri, d * apq);                    // sgn(d a_pq) |a_pq| / r
    const double t = q * (rw * rw);
    c = rot ? w * rw : 1.0;
    s = rot ? q * rw : 0.0;
    npp = rot ? app - t * apq : app;
    nqq = rot ? aqq + t * apq : aqq;
    return rot;
}

// the same criterion as a pure test (termination check)
__device__ __forceinline__ int would_rotate(double app, double aqq, double apq, double drop, double abs_floor, double conv2)
{
    const double aa = fabs(apq);
    const bool tiny = fmax(fmax(fabs(app), fabs(aqq)), aa) < TINY_FRACTION * drop;
    return (aa * aa > conv2 * fabs(app * aqq) && aa > abs_floor && !tiny) ? 1 : 0;
}

// One inner round inside a symmetric 4x4 block held in registers (both triangles kept): the two disjoint pairs (P1, Q1)
// and (P2, Q2) - their parameters come from the block as it stands (a rotation of one pair does not touch the three
// elements that define the other), then both two-sided rotations are applied, to the block and to the matching entries
// of the right-hand side.  Unrotated pairs carry (c, s) = (1, 0) through the same arithmetic (exact).
template <int P1, int Q1, int P2, int Q2>
__device__ __forceinline__ void rot_stage(double (&d)[4][4], double (&yy)[4], double drop, double abs_floor, int& rotated,
                                          double2& r1, double2& r2)
{
    double c1, s1, npp1, nqq1, c2, s2, npp2, nqq2;
    const bool g1 = rot_params(d[P1][P1], d[Q1][Q1], d[P1][Q1], drop, abs_floor, c1, s1, npp1, nqq1);
    const bool g2 = rot_params(d[P2][P2], d[Q2][Q2], d[P2][Q2], drop, abs_floor, c2, s2, npp2, nqq2);
    rotated |= (g1 || g2) ? 1 : 0;
    // the 2x2 cross block between the pairs: B = [[d(P1,P2), d(P1,Q2)], [d(Q1,P2), d(Q1,Q2)]] <- R1^T B R2
    const double b00 = d[P1][P2], b01 = d[P1][Q2], b10 = d[Q1][P2], b11 = d[Q1][Q2];
    const double t00 = c1 * b00 - s1 * b10, t01 = c1 * b01 - s1 * b11;
    const double t10 = s1 * b00 + c1 * b10, t11 = s1 * b01 + c1 * b11;
    d[P1][P2] = d[P2][P1] = c2 * t00 - s2 * t01;
    d[P1][Q2] = d[Q2][P1] = s2 * t00 + c2 * t01;
    d[Q1][P2] = d[P2][Q1] = c2 * t10 - s2 * t11;
    d[Q1][Q2] = d[Q2][Q1] = s2 * t10 + c2 * t11;
    d[P1][P1] = npp1;
    d[Q1][Q1] = nqq1;
    d[P2][P2] = npp2;
    d[Q2][Q2] = nqq2;
    d[P1][Q1] = d[Q1][P1] = g1 ? 0.0 : d[P1][Q1];
    d[P2][Q2] = d[Q2][P2] = g2 ? 0.0 : d[P2][Q2];
    const double y1p = yy[P1], y1q = yy[Q1], y2p = yy[P2], y2q = yy[Q2];
    yy[P1] = c1 * y1p - s1 * y1q;                    // y <- J^T y
    yy[Q1] = s1 * y1p + c1 * y1q;
    yy[P2] = c2 * y2p - s2 * y2q;
    yy[Q2] = s2 * y2p + c2 * y2q;
    r1 = make_double2(c1, s1);
    r2 = make_double2(c2, s2);
}

// rows P, Q of a 4x4 super-block <- R^T (rows P, Q):  row_P' = c row_P - s row_Q,  row_Q' = s row_P + c row_Q
template <int P, int Q>
__device__ __forceinline__ void rot_rows(double (&b)[16], double2 r)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double x = b[4 * P + k], z = b[4 * Q + k];
        b[4 * P + k] = r.x * x - r.y * z;
        b[4 * Q + k] = r.y * x + r.x * z;
    }
}
// columns P, Q of a 4x4 super-block <- (columns P, Q) R
template <int P, int Q>
__device__ __forceinline__ void rot_cols(double (&b)[16], double2 r)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double x = b[4 * k + P], z = b[4 * k + Q];
        b[4 * k + P] = r.x * x - r.y * z;
        b[4 * k + Q] = r.y * x + r.x * z;
    }
}

// ---- reverse replay of the rotation log on ONE vector held in the registers of one wave ------------------------
// g <- J g per round, where a round first rotated every match and then (rounds > 0 of a sweep) moved the units.  Lane a
// holds the four entries of match a (M <= 64).  In the reverse direction match a gathers its unit U from ring place
// next(2a) and its unit V from next(2a + 1): the U of the NEXT match and the V of the PREVIOUS one, with the exceptions
// a = 0 (own U; V <- U of match 1) and a = M - 1 (U <- own V), i.e. two single-lane shifts of the wave: DPP
// wave_shl:1 / wave_shr:1 (8 cycles each) instead of an LDS write + read per round.
struct WaveReplay {
    int lane, last;
    double x0, x1, x2, x3;          // U0, U1, V0, V1 of match `lane`

    __device__ __forceinline__ static double from_next(double v)     // lane i <- v[i + 1]
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ static double from_prev(double v)     // lane i <- v[i - 1]
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ void init(int lane_, int M)
    {
        lane = lane_;
        last = M - 1;
    }
    __device__ __forceinline__ void load(const double* y, int M)
    {
        const bool has = lane < M;
        x0 = has ? y[4 * lane] : 0.0;
        x1 = has ? y[4 * lane + 1] : 0.0;
        x2 = has ? y[4 * lane + 2] : 0.0;
        x3 = has ? y[4 * lane + 3] : 0.0;
    }
    __device__ __forceinline__ void store(double* y, int M) const
    {
        if (lane < M) { y[4 * lane] = x0; y[4 * lane + 1] = x1; y[4 * lane + 2] = x2; y[4 * lane + 3] = x3; }
    }
    __device__ __forceinline__ static void inv(double& p, double& q, double2 r)      // (p, q) <- J (p, q)
    {
        // the contractions are spelled out: left to the compiler, the 1- and 8-column instances of the eigenvector kernel
        // fused different products and a record's eigenvectors depended on the size of the batch it was solved in
        const double gp = p, gq = q;
        p = fma(r.x, gp, r.y * gq);
        q = fma(r.x, gq, -(r.y * gp));
    }
    // r[0..3]: (c, s) of this lane's match in the round ((1, 0) where there is no match); intra: round 0 of a sweep
    __device__ __forceinline__ void round(const double2 (&r)[4], bool intra)
    {
        if (intra) {                                   // pairs (U0,U1), (V0,V1); no permutation
            inv(x0, x1, r[0]);
            inv(x2, x3, r[1]);
            return;
        }
        const double n0 = from_next(x0), n1 = from_next(x1);     // U of the next match
        const double p2 = from_prev(x2), p3 = from_prev(x3);     // V of the previous match
        const double u0 = lane == 0 ? x0 : (lane == last ? x2 : n0);
        const double u1 = lane == 0 ? x1 : (lane == last ? x3 : n1);
        const double v0 = lane == 0 ? n0 : p2;
        const double v1 = lane == 0 ? n1 : p3;
        x0 = u0; x1 = u1; x2 = v0; x3 = v1;
        inv(x0, x3, r[2]);                             // inner round 2: (U0,V1), (U1,V0)
        inv(x1, x2, r[3]);
        inv(x0, x2, r[0]);                             // inner round 1: (U0,V0), (U1,V1)
        inv(x1, x3, r[1]);
    }
};

// One workgroup per system; blockDim.x = NT threads (a multiple of 64), IT super-blocks per thread.
//
// Where a round spends its cycles (tools/exp_stamps.py, N = 144, 640 threads): 4500 cycles, of which 1800 are the
// rotation set-up - a chain of ~60 dependent fp64 operations in the first M lanes of wave 0 while the other waves wait
// at the barrier - and 1500-2400 the block updates (128 fp64 operations and 16 + 16 LDS accesses per thread, three waves
// per SIMD).  The set-up was 2700 cycles with library sqrt / division and the two rotations of an inner round under
// separate branches; it is now division-free and branch-free (rot_params, rot_stage).  (Skipping the chain and the block
// arithmetic in rounds where no pair rotates - most rounds of the late sweeps - was measured too: the extra control flow
// cost registers (16 spilled) and the kernel ran 163 us per sweep instead of 150.)  Two ways of taking it off the
// critical path were built and measured, and both lost: computing the next round's rotations one round ahead in a
// dedicated wave (the diagonal blocks assembled from registers of the designated super-blocks: 170-260 us per sweep
// against 154, the extra state spilled to scratch under the 168-register budget of three waves per SIMD), and letting
// wave 0 run ahead into the next round's set-up right after its own stores (176 us: its LDS round trip queues behind
// the other waves' block stores, the chain took 3800 cycles instead of 1800; raising its priority changed nothing).
template <int IT>
__device__ __forceinline__ void jacobi_system(
    unsigned char* lds_raw, int N, const double* __restrict__ Xs, double sc, const double* __restrict__ ys, double rcond,
    double abs_floor, double* __restrict__ Cs, int* __restrict__ rank_s, double2* __restrict__ logp, int max_sweeps,
    int* __restrict__ sweeps_s, double* __restrict__ lam_s, int lam_raw, int* __restrict__ nround_s,
    unsigned long long* __restrict__ round_acc, double conv_tol = 0.0)
{
    const int NT = blockDim.x;
    const int Np = (N + 3) & ~3;          // padded to a multiple of four with inert indices
    const int m = Np >> 1;                // units
    const int M = Np >> 2;                // matches
    const int nsb = (M * (M - 1)) >> 1;
    const int ntri = 16 * nsb + 10 * M;   // = Np (Np + 1) / 2
    const int dg = 16 * nsb;              // diagonal planes
    double* A = reinterpret_cast<double*>(lds_raw);                          // [ntri], slot-indexed (even count)
    double* yv = A + ntri;                                                   // [2][Np] double-buffered
    double2* cs = reinterpret_cast<double2*>(yv + 2 * Np);                   // [4][M], 16-B aligned
    double* nd = reinterpret_cast<double*>(cs + 4 * M);                      // [16] reduction scratch

    const int tid = threadIdx.x;
    const int nw = NT >> 6;

    // my super-blocks (a < b), fixed for the whole run: k = b(b-1)/2 + a; source / destination addresses
    int ka[IT], kb[IT], ksrc[IT], dst[IT][16];
    bool live[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int ko = tid + it * NT;
        live[it] = ko < nsb;
        const int k = live[it] ? ko : 0;
        int b = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)k)) * 0.5f);
        while ((b * (b - 1)) / 2 > k) --b;
        while (((b + 1) * b) / 2 <= k) ++b;
        const int a = k - (b * (b - 1)) / 2;
        ka[it] = live[it] ? a : 0;
        kb[it] = live[it] ? b : 1;
        ksrc[it] = k;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                dst[it][4 * r + c] = tri4(slot_next(4 * ka[it] + r, M), slot_next(4 * kb[it] + c, M), M);
    }
    // diagonal block of my match (threads < M): source planes dg + j * M + tid, permuted destinations
    const int da = tid < M ? tid : 0;
    int ddst[10], ydst[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ydst[r] = slot_next(4 * da + r, M);
#pragma unroll
        for (int c = r; c < 4; ++c) ddst[j10(r, c)] = tri4(slot_next(4 * da + r, M), slot_next(4 * da + c, M), M);
    }

    // ---- load (slot s holds original index s) ---------------------------------------------------------
    double mxd = 0.0;
    for (int e = tid; e < Np * Np; e += NT) {
        const int i = e / Np, j = e - i * Np;
        if (j > i) continue;
        const double v = (i < N && j < N) ? Xs[(int64_t)i * N + j] : 0.0;
        A[tri4(i, j, M)] = v;
        if (i == j) mxd = fmax(mxd, fabs(v));
    }
    for (int s = tid; s < Np; s += NT) yv[s] = s < N ? ys[s] : 0.0;
    for (int o = 32; o > 0; o >>= 1) mxd = fmax(mxd, __shfl_xor(mxd, o));
    if ((tid & 63) == 0) nd[tid >> 6] = mxd;
    __syncthreads();
    mxd = 0.0;
    for (int w = 0; w < nw; ++w) mxd = fmax(mxd, nd[w]);
    // pairs lying wholly inside the to-be-truncated subspace need no further work: the cut is
    // rcond * max|lambda| and max|diag| <= max|lambda|, so this never skips a pair that reaches the cut
    const double drop = rcond * mxd;
    const double eps2 = 2.220446049250313e-16 * 2.220446049250313e-16;
    // termination test, see below; conv_tol > 0: the iteration ends once every |a_pq| <= conv_tol sqrt|a_pp a_qq| (the solves of
    // the bracket walk, whose chi^2 only decides signs: vi_basis_solve_f64)
    const double conv2 = conv_tol > 0.0 ? conv_tol * conv_tol : VI_CONV_FACTOR * VI_CONV_FACTOR * eps2;
    int sweep = 0, ycur = 0;
    bool converged = false;               // false: the sweep cap ended the iteration
    int64_t nround = 0;
#ifdef VI_STAMPS
    unsigned long long stamp_t = __builtin_readcyclecounter();
#endif
    for (; sweep < max_sweeps; ++sweep) {
        int rotated = 0;
        VI_ROT_SWEEP(sweep);
        for (int r = 0; r < m; ++r, ++nround) {
            const bool intra = r == 0;            // round 0 of a sweep: the pairs inside the units, no permutation
            const double2* csc = cs;
            VI_STAMP(7);
            // ---- phase 1 (first M threads): the rotations of every match, both inner rounds ---------------------
            double d[4][4];
            if (tid < M) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = p; q < 4; ++q) d[p][q] = d[q][p] = A[dg + j10(p, q) * M + tid];
                double yy[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) yy[p] = yv[ycur * Np + 4 * tid + p];
                double2 r0, r1, r2 = make_double2(1.0, 0.0), r3 = make_double2(1.0, 0.0);
                if (intra) {
                    rot_stage<0, 1, 2, 3>(d, yy, drop, abs_floor, rotated, r0, r1);
                } else {
                    rot_stage<0, 2, 1, 3>(d, yy, drop, abs_floor, rotated, r0, r1);
                    rot_stage<0, 3, 1, 2>(d, yy, drop, abs_floor, rotated, r2, r3);
                }
                VI_ROT_COUNT(sweep, r0, r1, r2, r3);
                cs[tid] = r0;
                cs[M + tid] = r1;
                cs[2 * M + tid] = r2;
                cs[3 * M + tid] = r3;
                double2* lp = logp + nround * (int64_t)(4 * M) + tid;
                lp[0] = r0;
                lp[M] = r1;
                lp[2 * M] = r2;
                lp[3 * M] = r3;
#pragma unroll
                for (int p = 0; p < 4; ++p) yv[(ycur ^ 1) * Np + (intra ? 4 * tid + p : ydst[p])] = yy[p];
            }
            VI_STAMP(0);
            // ---- phase 2a (all threads, overlaps phase 1 of wave 0): fetch my super-blocks ---------------------------------------------------------
            double b[IT][16];
#pragma unroll
            for (int it = 0; it < IT; ++it)
                if (live[it]) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) b[it][e] = A[ksrc[it] + e * nsb];
                }
            VI_STAMP(1);
            __syncthreads();
            VI_STAMP(2);
            // ---- phase b: S_ab <- R_a^T S_ab R_b, stored at the permuted slots ----------------------------------
#pragma unroll
            for (int it = 0; it < IT; ++it)
                if (live[it]) {
                    const double2 ra0 = csc[ka[it]], ra1 = csc[M + ka[it]];
                    const double2 rb0 = csc[kb[it]], rb1 = csc[M + kb[it]];
                    if (intra) {
                        rot_rows<0, 1>(b[it], ra0);
                        rot_rows<2, 3>(b[it], ra1);
                        rot_cols<0, 1>(b[it], rb0);
                        rot_cols<2, 3>(b[it], rb1);
#pragma unroll
                        for (int e = 0; e < 16; ++e) A[ksrc[it] + e * nsb] = b[it][e];
                    } else {
                        const double2 ra2 = csc[2 * M + ka[it]], ra3 = csc[3 * M + ka[it]];
                        const double2 rb2 = csc[2 * M + kb[it]], rb3 = csc[3 * M + kb[it]];
                        rot_rows<0, 2>(b[it], ra0);
                        rot_rows<1, 3>(b[it], ra1);
                        rot_rows<0, 3>(b[it], ra2);
                        rot_rows<1, 2>(b[it], ra3);
                        rot_cols<0, 2>(b[it], rb0);
                        rot_cols<1, 3>(b[it], rb1);
                        rot_cols<0, 3>(b[it], rb2);
                        rot_cols<1, 2>(b[it], rb3);
#pragma unroll
                        for (int e = 0; e < 16; ++e) A[dst[it][e]] = b[it][e];
                    }
                }
            if (tid < M) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = p; q < 4; ++q)
                        A[intra ? dg + j10(p, q) * M + tid : ddst[j10(p, q)]] = d[p][q];
            }
            ycur ^= 1;
            VI_STAMP(3);
            __syncthreads();
            VI_STAMP(4);
        }
        // The classical iteration ends with a sweep that rotates nothing - m rounds of pure data movement.  That sweep
        // applies the rotation criterion to every pair of an unchanged matrix, so its outcome is known now: test all
        // pairs in place (each thread its own blocks) and stop if none would rotate.  The slot arrangement after a
        // whole sweep is the initial one.
        if (!__syncthreads_or(rotated)) { ++sweep; converged = true; break; }
        {
            int viol = 0;
#pragma unroll
            for (int it = 0; it < IT; ++it)
                if (live[it]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double app = A[dg + r * M + ka[it]];
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            viol |= would_rotate(app, A[dg + c * M + kb[it]], A[ksrc[it] + (4 * r + c) * nsb], drop,
                                                 abs_floor, conv2);
                    }
                }
            if (tid < M) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = p + 1; q < 4; ++q)
                        viol |= would_rotate(A[dg + p * M + tid], A[dg + q * M + tid], A[dg + j10(p, q) * M + tid], drop,
                                             abs_floor, conv2);
            }
            if (!__syncthreads_or(viol)) { ++sweep; converged = true; break; }
        }
    }
    // ---- truncated solve in the eigenbasis (slot order = original order) ------------------------------
    double* yc = yv + ycur * Np;
    double mx = 0.0;
    for (int i = tid; i < Np; i += NT) mx = fmax(mx, fabs(A[dg + (i & 3) * M + (i >> 2)]));
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) nd[tid >> 6] = mx;
    __syncthreads();
    mx = 0.0;
    for (int w = 0; w < nw; ++w) mx = fmax(mx, nd[w]);
    const double thr = rcond * mx;
    int rk = 0;
    for (int i = tid; i < Np; i += NT) {
        const double lam = A[dg + (i & 3) * M + (i >> 2)];
        const bool keep = fabs(lam) > thr;
        yc[i] = keep ? yc[i] / (lam * sc) : 0.0;
        rk += keep ? 1 : 0;
        // a padding index is an exact zero eigenvalue and is never kept; eigenvalues leave unsorted
        // (with lam_raw the eigenvalues of the scaled system, as k_trunc_apply expects)
        if (lam_s && i < N) lam_s[i] = lam_raw ? lam : lam * sc;
    }
    __syncthreads();                            // nd is reused below
    for (int o = 32; o > 0; o >>= 1) rk += __shfl_xor(rk, o);
    if ((tid & 63) == 0) nd[tid >> 6] = (double)rk;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < nw; ++w) tot += nd[w];
        if (rank_s) *rank_s = (int)tot;
        if (sweeps_s) *sweeps_s = converged ? sweep : max_sweeps + 1;      // cap + 1: not converged
        if (nround_s) *nround_s = (int)nround;
        if (round_acc) atomicAdd(round_acc, (unsigned long long)nround);      // bench only (vi_solve_timing)
    }
    // ---- C = V g : undo (permutation, rotations) round by round -----------------------------------------------
    // Done by wave 0 alone with the vector in its registers (WaveReplay): a workgroup barrier per round would cost
    // more than the round itself.
    if (tid < 64) {
        constexpr int PF = 4;                           // rounds of (c, s) prefetched per batch
        WaveReplay W;
        W.init(tid, M);
        W.load(yc, M);
        const bool has = tid < M;
        for (int64_t r1 = nround; r1 > 0; r1 -= PF) {
            const int nb = r1 >= PF ? PF : (int)r1;
            double2 pf[PF][4];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) pf[u][j] = make_double2(1.0, 0.0);
                if (u < nb && has) {
                    const double2* lp = logp + (r1 - 1 - u) * (int64_t)(4 * M) + tid;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pf[u][j] = lp[j * M];
                }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (u < nb) W.round(pf[u], ((r1 - 1 - u) % m) == 0);
        }
        W.store(yc, M);                                 // back into the buffer it came from
    }
    __syncthreads();
    for (int s = tid; s < N; s += NT) Cs[s] = yc[s];
}

// Eigenvectors CPW at a time by ONE wave (all 64 lanes call it): columns col0 .. col0 + CPW - 1 of V = J_1 J_2 ... J_K from the
// rotation log of a system, written to Vs (LAPACK layout: eigenvector k at Vs[k * N ..]).
template <int CPW>
__device__ __forceinline__ void jacobi_vector_strip(int N, const double2* __restrict__ logp, int64_t nround, int col0, int lane,
                                                    double* __restrict__ Vs)
{
    const int Np = (N + 3) & ~3, m = Np >> 1, M = Np >> 2;
    WaveReplay W[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int col = col0 + c;
        W[c].init(lane, M);
        W[c].x0 = (4 * lane == col) ? 1.0 : 0.0;
        W[c].x1 = (4 * lane + 1 == col) ? 1.0 : 0.0;
        W[c].x2 = (4 * lane + 2 == col) ? 1.0 : 0.0;
        W[c].x3 = (4 * lane + 3 == col) ? 1.0 : 0.0;
    }
    constexpr int PF = 4;                               // rounds of (c, s) prefetched per batch
    const bool has = lane < M;
    for (int64_t r1 = nround; r1 > 0; r1 -= PF) {
        const int nb = r1 >= PF ? PF : (int)r1;
        double2 pf[PF][4];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pf[u][j] = make_double2(1.0, 0.0);
            if (u < nb && has) {
                const double2* lp = logp + (r1 - 1 - u) * (int64_t)(4 * M) + lane;
#pragma unroll
                for (int j = 0; j < 4; ++j) pf[u][j] = lp[j * M];
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (u < nb) {
                const bool intra = ((r1 - 1 - u) % m) == 0;
#pragma unroll
                for (int c = 0; c < CPW; ++c) W[c].round(pf[u], intra);
            }
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int col = col0 + c;
        if (col >= N) break;
        double* Vo = Vs + (int64_t)col * N;                              // eigenvector `col`, indexed by original index
        if (has) {
            if (4 * lane < N) Vo[4 * lane] = W[c].x0;
            if (4 * lane + 1 < N) Vo[4 * lane + 1] = W[c].x1;
            if (4 * lane + 2 < N) Vo[4 * lane + 2] = W[c].x2;
            if (4 * lane + 3 < N) Vo[4 * lane + 3] = W[c].x3;
        }
    }
}


}  // namespace
