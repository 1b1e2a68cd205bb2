// K3: batched truncated minimum-norm solve of small symmetric (indefinite, possibly rank-deficient)
// systems by a parallel cyclic Jacobi eigensolver that lives entirely in one CU's LDS.
//
// Replaces, for the hundreds of trial systems per record of the regularisation-parameter search,
// scipy.linalg.lstsq (LAPACK gelsd, rcond = eps) at volumetricinterp/interpolate.py:462:
// for symmetric X = V diag(lam) V^T the SVD truncation |sigma| <= rcond*sigma_max is the spectral
// truncation |lam| <= rcond*max|lam|, and the minimum-norm solution is V diag(1/lam | kept) V^T y.
//
// One 512-thread workgroup per system.  The symmetric half of X (N(N+1)/2 doubles: 83.5 KB at
// N = 144, inside the 160 KB LDS of a gfx950 CU) stays in LDS for the whole decomposition.
//  * Round-robin ordering with DATA movement instead of index movement: the matrix is stored by "slot";
//    pair P is always slots (2P, 2P+1), and after every round the Brent-Luk "music chairs" permutation pi
//    is applied to the slots.  Because a round reads every 2x2 block B_PQ (rows of pair P, columns of pair
//    Q) into registers before the barrier and writes it after, the permutation is free: B'_PQ is simply
//    stored at (pi(row), pi(col)).  All LDS addresses of a thread are therefore round-invariant and live in
//    registers - the inner loop has no index arithmetic at all.
//  * The two-sided update A <- J^T A J with J the direct sum of the m = N/2 disjoint plane rotations of a
//    round factorises into independent updates B_PQ <- R_P^T B_PQ R_Q: no atomics, two barriers per round,
//    and the block fetch of the other waves overlaps the (sqrt/div-latency-bound) rotation set-up.
//  * The eigenvector matrix is never formed: the right-hand side is rotated along (y <- J^T y), the
//    rotations (c, s) are logged to HBM (16 B each, ~1.3 MB per system at N = 144) and replayed in reverse
//    on the truncated, scaled vector to give C = V g.
//  * Convergence: a pair is rotated while |a_pq| > eps*sqrt(|a_pp a_qq|) (relative criterion: small kept
//    eigenvalues stay accurate), except below an absolute floor and inside the subspace that the
//    truncation is going to drop anyway; a sweep without rotations ends the iteration.
#include "vi_common.h"

#include <cstdlib>
#include <cstring>

#ifdef VI_STAMPS
// diagnostic build only (make STAMPS=1): per-phase cycle sums of workgroup 0, wave 0
__device__ unsigned long long g_jacobi_stamps[8];
__device__ int g_jacobi_rot[64];       // rotations per sweep of workgroup 0 in the last launch
#define VI_STAMP(k)                                                                   \
    do {                                                                              \
        const unsigned long long t_ = __builtin_readcyclecounter();                   \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_jacobi_stamps[k] += t_ - stamp_t;  \
        stamp_t = __builtin_readcyclecounter();                                       \
    } while (0)
#else
#define VI_STAMP(k)
#endif

namespace {

// Termination tolerance of the all-pairs test in units of the rotation criterion (eps).  1 = the classical rule.  A
// looser test (4 eps, like xGESVJ's sqrt(N) eps) was measured and ends no solve earlier: the late sweeps of a cold solve
// (tools/exp_rotcounts.py: 279, 145, 95, 66, 58, 35, 34, 35, 34, ... rotations per sweep up to the cap of 24) chase
// absolute rounding noise between tiny diagonal entries, far above any relative tolerance.
#ifndef VI_CONV_FACTOR
#define VI_CONV_FACTOR 1.0
#endif
constexpr int JBS = 512;              // 8 waves: two per SIMD, so LDS latency of one hides under the other

// LDS address of element (i, j) of the slot-indexed symmetric matrix, m = number of pairs.  The m(m-1)/2
// off-diagonal 2x2 blocks B_PQ (P < Q, block index k = Q(Q-1)/2 + P, the order the threads enumerate them in)
// are stored as four planes E[e][k], e = 2a + b for element (2P+a, 2Q+b); the diagonal blocks as three arrays
// (a_pp, a_qq, a_pq).  Consecutive lanes own consecutive blocks, so the four block reads of a round are
// perfectly linear (bank-conflict free; a packed-triangle layout measured 32 % conflict cycles) and the
// permuted stores stay linear within a block column.
__device__ __host__ __forceinline__ int tri_m(int i, int j, int m)
{
    const int p = i >> 1, a = i & 1, q = j >> 1, b = j & 1;
    const int nblk = (m * (m - 1)) >> 1;
    if (p == q) return 4 * nblk + (a == b ? (a ? m + p : p) : 2 * m + p);
    if (p < q) return (2 * a + b) * nblk + ((q * (q - 1)) >> 1) + p;
    return (2 * b + a) * nblk + ((p * (p - 1)) >> 1) + q;
}
#define tri(i, j) tri_m((i), (j), m)

// slot permutation applied after every round (Brent & Luk): slot 0 is fixed, the others advance along the
// ring  1 -> 2 -> 4 -> ... -> 2m-2 -> 2m-1 -> 2m-3 -> ... -> 3 -> 1
__device__ __forceinline__ int slot_next(int s, int m)
{
    if (s == 0) return 0;
    if (s == 1) return 2;
    if (s & 1) return s - 2;
    return s == 2 * m - 2 ? 2 * m - 1 : s + 2;
}
// original index held by slot s in the initial arrangement: top row 0..m-1, bottom row m..2m-1
__device__ __forceinline__ int slot_orig0(int s, int m) { return (s & 1) ? m + (s >> 1) : (s >> 1); }

// ---- reverse replay of the rotation log on ONE vector held in the registers of one wave ------------------------
// g <- J g per round, where round r first rotated every pair and then applied the slot permutation pi
// (slot_next).  In the reverse direction pair P = (slots 2P, 2P+1) therefore gathers from pi(2P), pi(2P+1): the
// TOP element of the next pair and the BOTTOM element of the previous one (the music-chairs ring), with two
// exceptions: pair 0 takes (top[0], top[1]) and the last pair takes (its own bottom, bot[m-2]).  Lane l holds pairs
// l and l + 64 (m <= 128), so both gathers are single-lane shifts of the wave: DPP wave_shl:1 / wave_shr:1 (8 cycles)
// instead of an LDS write + read per round (the LDS version spent ~350 cycles per round on that round trip, 19 %
// of the whole solve; tools/exp_stamps.py).  Arithmetic and its order are those of the LDS version.
struct WaveReplay {
    int lane, last0, last1;
    bool has0, has1;
    double t0, b0, t1, b1;

    __device__ __forceinline__ static double from_next(double v, double edge)    // lane i <- v[i + 1]; lane 63 <- edge
    {
        const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ static double from_prev(double v, double edge)    // lane i <- v[i - 1]; lane 0 <- edge
    {
        const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ static double lane_value(double v, int l)          // wave-uniform copy of v[l]
    {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                                __builtin_amdgcn_readlane(__double2loint(v), l));
    }
    __device__ __forceinline__ void init(int lane_, int m)
    {
        lane = lane_;
        has0 = lane < m;
        has1 = lane + 64 < m;
        last0 = m <= 64 ? m - 1 : -1;
        last1 = m > 64 ? m - 1 - 64 : -1;
    }
    __device__ __forceinline__ void load(const double* y)
    {
        t0 = has0 ? y[2 * lane] : 0.0;
        b0 = has0 ? y[2 * lane + 1] : 0.0;
        t1 = has1 ? y[2 * (lane + 64)] : 0.0;
        b1 = has1 ? y[2 * (lane + 64) + 1] : 0.0;
    }
    __device__ __forceinline__ void store(double* y) const
    {
        if (has0) { y[2 * lane] = t0; y[2 * lane + 1] = b0; }
        if (has1) { y[2 * (lane + 64)] = t1; y[2 * (lane + 64) + 1] = b1; }
    }
    // r0 / r1: (c, s) of pairs lane / lane + 64 in this round ((1, 0) where there is no pair)
    __device__ __forceinline__ void round(double2 r0, double2 r1)
    {
        const double top64 = lane_value(t1, 0);          // top[64]  -> gather of pair 63
        const double bot63 = lane_value(b0, 63);         // bot[63]  -> gather of pair 64
        const double nx0 = from_next(t0, top64);         // top[l + 1]
        const double nx1 = from_next(t1, 0.0);           // top[l + 65]
        const double pv0 = from_prev(b0, 0.0);           // bot[l - 1]
        const double pv1 = from_prev(b1, bot63);         // bot[l + 63]
        const double gp0 = lane == 0 ? t0 : (lane == last0 ? b0 : nx0);
        const double gq0 = lane == 0 ? nx0 : pv0;
        const double gp1 = lane == last1 ? b1 : nx1;
        const double gq1 = pv1;
        t0 = r0.x * gp0 + r0.y * gq0;
        b0 = -r0.y * gp0 + r0.x * gq0;
        t1 = r1.x * gp1 + r1.y * gq1;
        b1 = -r1.y * gp1 + r1.x * gq1;
    }
};

template <int IT>
__global__ __launch_bounds__(JBS) void k_jacobi_solve(int N, const double* __restrict__ X,
                                                      const double* __restrict__ scl, const double* __restrict__ y,
                                                      const int* __restrict__ rec, double rcond, double abs_floor,
                                                      double* __restrict__ C, int* __restrict__ rank,
                                                      double2* __restrict__ rotlog, int64_t log_stride,
                                                      int max_sweeps, int* __restrict__ sweeps_out,
                                                      double* __restrict__ lam_out, int lam_raw,
                                                      int* __restrict__ nround_out,
                                                      unsigned long long* __restrict__ round_acc)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int Np = (N + 1) & ~1;          // padded to even with an inert index
    const int m = Np >> 1;
    const int ntri = (Np * (Np + 1)) >> 1;
    const int nblk = (m * (m - 1)) >> 1;
    double* A = reinterpret_cast<double*>(lds_raw);                          // [ntri], slot-indexed
    double* yv = A + ntri;                                                   // [2][Np] double-buffered
    double2* cs = reinterpret_cast<double2*>(yv + 2 * Np + (ntri & 1));      // [m], 16-B aligned
    double* nd = reinterpret_cast<double*>(cs + m);                          // [8] reduction scratch
    const int trash0 = (int)((nd + 8) - A);                                  // [4 (IT JBS - nblk)] idle block slots

    const int tid = threadIdx.x;
    const int64_t sys = blockIdx.x;
    const double* Xs = X + sys * (int64_t)N * N;
    const double* ys = y + (int64_t)(rec ? rec[sys] : sys) * N;
    double2* logp = rotlog + sys * log_stride;

    // ---- load (slot s holds original index slot_orig0(s)) ----------------------------------------
    double mxd = 0.0;
    for (int e = tid; e < ntri; e += JBS) {
        // e -> (row, col) of the lower triangle (enumeration order only; the LDS position is tri(row, col))
        int row = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while (((row + 1) * (row + 2)) / 2 <= e) ++row;
        while ((row * (row + 1)) / 2 > e) --row;
        const int col = e - (row * (row + 1)) / 2;
        const int oi = slot_orig0(row, m), oj = slot_orig0(col, m);
        const double v = (oi < N && oj < N) ? Xs[(int64_t)oi * N + oj] : 0.0;
        A[tri(row, col)] = v;
        if (row == col) mxd = fmax(mxd, fabs(v));
    }
    for (int s = tid; s < Np; s += JBS) {
        const int o = slot_orig0(s, m);
        yv[s] = o < N ? ys[o] : 0.0;
    }
    for (int o = 32; o > 0; o >>= 1) mxd = fmax(mxd, __shfl_xor(mxd, o));
    if ((tid & 63) == 0) nd[tid >> 6] = mxd;
    // my 2x2 blocks (P < Q), fixed for the whole run: k = Q(Q-1)/2 + P; source / destination addresses
    int bP[IT], bQ[IT], src[IT][4], dst[IT][4];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int k = tid + it * JBS;
        int Q = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)k)) * 0.5f);
        while ((Q * (Q - 1)) / 2 > k) --Q;
        while (((Q + 1) * Q) / 2 <= k) ++Q;
        const int P = k - (Q * (Q - 1)) / 2;
        const bool live = k < nblk;
        bQ[it] = live ? Q : 0;
        bP[it] = live ? P : 0;
        // element (a, b) of B_PQ = A(slot 2P+a, slot 2Q+b); idle block slots of the last iteration work on a
        // private scratch quadruple so that the loop body needs no predication
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                src[it][2 * a + b2] = live ? tri(2 * P + a, 2 * Q + b2) : trash0 + 4 * (k - nblk) + 2 * a + b2;
                dst[it][2 * a + b2] = live ? tri(slot_next(2 * P + a, m), slot_next(2 * Q + b2, m))
                                           : trash0 + 4 * (k - nblk) + 2 * a + b2;
            }
        if (!live)
            for (int e = 0; e < 4; ++e) A[src[it][e]] = 0.0;
    }
    // diagonal block of my pair (threads < m)
    const int p0 = 2 * tid, p1 = 2 * tid + 1;
    const int n0 = slot_next(p0 < Np ? p0 : 0, m), n1 = slot_next(p1 < Np ? p1 : 1, m);
    const int spp = tri(p0, p0), sqq = tri(p1, p1), spq = tri(p1, p0);
    const int dpp = tri(n0, n0), dqq = tri(n1, n1), dpq = tri(n0, n1);
    __syncthreads();
    mxd = 0.0;
#pragma unroll
    for (int w = 0; w < JBS / 64; ++w) mxd = fmax(mxd, nd[w]);
    // pairs lying wholly inside the to-be-truncated subspace need no further work: the cut is
    // rcond * max|lambda| and max|diag| <= max|lambda|, so this never skips a pair that reaches the cut
    const double drop = rcond * mxd;
    const double eps2 = 2.220446049250313e-16 * 2.220446049250313e-16;
    const double conv2 = VI_CONV_FACTOR * VI_CONV_FACTOR * eps2;      // termination test, see below
    const int rps = 2 * m - 1;            // rounds per sweep
    int sweep = 0, ycur = 0;
    int64_t nround = 0;
#ifdef VI_STAMPS
    unsigned long long stamp_t = __builtin_readcyclecounter();
#endif
    for (; sweep < max_sweeps; ++sweep) {
        int rotated = 0;
#ifdef VI_STAMPS
        int rot_sweep = 0;
#endif
        for (int r = 0; r < rps; ++r, ++nround) {
            VI_STAMP(7);
            // ---- phase 1 (first m threads): rotation of every pair ------------------------------------
            double npp = 0.0, nqq = 0.0, npq = 0.0;
            if (tid < m) {
                const double app = A[spp], aqq = A[sqq], apq = A[spq];
                double c = 1.0, s = 0.0;
                npp = app; nqq = aqq; npq = apq;
                const double aa = fabs(apq);
                const bool tiny = fmax(fmax(fabs(app), fabs(aqq)), aa) < drop;
                if (aa * aa > eps2 * fabs(app * aqq) && aa > abs_floor && !tiny) {
                    // t = tan(theta) = sgn(d) 2 a_pq / (|d| + sqrt(d^2 + 4 a_pq^2)),  d = a_qq - a_pp
                    const double d = aqq - app;
                    const double t = copysign(2.0 * apq, d * apq) / (fabs(d) + sqrt(fma(d, d, 4.0 * apq * apq)));
                    c = rsqrt(fma(t, t, 1.0));
                    s = t * c;
                    npp = app - t * apq;
                    nqq = aqq + t * apq;
                    npq = 0.0;
                    rotated = 1;
#ifdef VI_STAMPS
                    rot_sweep += 1;
#endif
                }
                const double yp = yv[ycur * Np + p0], yq = yv[ycur * Np + p1];
                yv[(ycur ^ 1) * Np + n0] = c * yp - s * yq;        // y <- J^T y, stored at the permuted slots
                yv[(ycur ^ 1) * Np + n1] = s * yp + c * yq;
                cs[tid] = make_double2(c, s);
                logp[nround * m + tid] = make_double2(c, s);
            }
            VI_STAMP(0);
            // ---- phase 2a (all threads, overlaps phase 1 of the other waves): fetch my blocks ------------
            double b[IT][4];
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[it][e] = A[src[it][e]];
            VI_STAMP(1);
            __syncthreads();
            VI_STAMP(2);
            // ---- phase 2b: B_PQ <- R_P^T B_PQ R_Q, stored at the permuted slots -----------------------------
            double2 rp[IT], rq[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                rp[it] = cs[bP[it]];
                rq[it] = cs[bQ[it]];
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const double t11 = rp[it].x * b[it][0] - rp[it].y * b[it][2], t12 = rp[it].x * b[it][1] - rp[it].y * b[it][3];
                const double t21 = rp[it].y * b[it][0] + rp[it].x * b[it][2], t22 = rp[it].y * b[it][1] + rp[it].x * b[it][3];
                A[dst[it][0]] = rq[it].x * t11 - rq[it].y * t12;
                A[dst[it][1]] = rq[it].y * t11 + rq[it].x * t12;
                A[dst[it][2]] = rq[it].x * t21 - rq[it].y * t22;
                A[dst[it][3]] = rq[it].y * t21 + rq[it].x * t22;
            }
            if (tid < m) {
                A[dpp] = npp;
                A[dqq] = nqq;
                A[dpq] = npq;
            }
            ycur ^= 1;
            VI_STAMP(3);
            __syncthreads();
            VI_STAMP(4);
        }
#ifdef VI_STAMPS
        {
            __shared__ int dbg_cnt;
            if (tid == 0) dbg_cnt = 0;
            __syncthreads();
            if (rot_sweep) atomicAdd(&dbg_cnt, rot_sweep);
            __syncthreads();
            if (blockIdx.x == 0 && tid == 0 && sweep < 64) g_jacobi_rot[sweep] = dbg_cnt;
            if (blockIdx.x == 0 && tid == 0 && sweep + 1 < 64) g_jacobi_rot[sweep + 1] = -1;
        }
#endif
        if (!__syncthreads_or(rotated)) { ++sweep; break; }
        // The iteration ends with a sweep that rotates nothing - 2m - 1 rounds of pure data movement (a quarter of a
        // warm solve).  That sweep applies the rotation criterion to every pair of an unchanged matrix, so its outcome
        // is known now: test all pairs in place (each thread its own blocks, ~2 us) and stop if none would rotate.  The
        // slot arrangement after a whole sweep is the initial one, so nothing else changes (results bit-identical).
        {
            int viol = 0;
            const int dg = 4 * nblk;                    // diagonal of slot 2P + a: A[dg + (a ? m + P : P)]
#pragma unroll
            for (int it = 0; it < IT; ++it) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2) {
                        const double apq = A[src[it][2 * a + b2]];
                        const double app = A[dg + (a ? m + bP[it] : bP[it])];
                        const double aqq = A[dg + (b2 ? m + bQ[it] : bQ[it])];
                        const double aa = fabs(apq);
                        const bool tiny = fmax(fmax(fabs(app), fabs(aqq)), aa) < drop;
                        viol |= (aa * aa > conv2 * fabs(app * aqq) && aa > abs_floor && !tiny) ? 1 : 0;
                    }
            }
            if (tid < m) {
                const double app = A[spp], aqq = A[sqq], apq = A[spq];
                const double aa = fabs(apq);
                const bool tiny = fmax(fmax(fabs(app), fabs(aqq)), aa) < drop;
                viol |= (aa * aa > conv2 * fabs(app * aqq) && aa > abs_floor && !tiny) ? 1 : 0;
            }
            if (!__syncthreads_or(viol)) { ++sweep; break; }
        }
    }
    // ---- truncated solve in the eigenbasis (slot order) ---------------------------------------------
    double* yc = yv + ycur * Np;
    double mx = 0.0;
    for (int i = tid; i < Np; i += JBS) mx = fmax(mx, fabs(A[tri(i, i)]));
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) nd[tid >> 6] = mx;
    __syncthreads();
    mx = 0.0;
#pragma unroll
    for (int w = 0; w < JBS / 64; ++w) mx = fmax(mx, nd[w]);
    const double thr = rcond * mx;
    const double sc = scl ? scl[sys] : 1.0;
    int rk = 0;
    for (int i = tid; i < Np; i += JBS) {
        const double lam = A[tri(i, i)];
        const bool keep = fabs(lam) > thr;
        yc[i] = keep ? yc[i] / (lam * sc) : 0.0;
        rk += keep ? 1 : 0;
        // the padding index is an exact zero eigenvalue and is never kept; eigenvalues leave unsorted
        // (slot order; with lam_raw the eigenvalues of the scaled system, as k_trunc_apply expects)
        if (lam_out && i < N) lam_out[sys * N + i] = lam_raw ? lam : lam * sc;
    }
    __syncthreads();                            // nd is reused below
    for (int o = 32; o > 0; o >>= 1) rk += __shfl_xor(rk, o);
    if ((tid & 63) == 0) nd[tid >> 6] = (double)rk;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < JBS / 64; ++w) tot += nd[w];
        if (rank) rank[sys] = (int)tot;
        if (sweeps_out) sweeps_out[sys] = sweep;
        if (nround_out) nround_out[sys] = (int)nround;
        if (round_acc) atomicAdd(round_acc, (unsigned long long)nround);      // bench only (vi_solve_timing)
    }
    VI_STAMP(5);
    // ---- C = V g : undo (permutation, rotation) round by round ----------------------------------------------
    // Done by wave 0 alone with the vector in its registers (WaveReplay): a workgroup barrier per round cost ~1.1k
    // cycles with 8 waves (31 % of the kernel), a single wave going through LDS ~350 cycles (19 %).
    if (tid < 64) {
        constexpr int PF = 8;                           // rounds of (c, s) prefetched per batch
        const int P1 = tid + 64;
        WaveReplay W;
        W.init(tid, m);
        W.load(yc);
        for (int64_t r1 = nround; r1 > 0; r1 -= PF) {
            const int nb = r1 >= PF ? PF : (int)r1;
            double2 pf0[PF], pf1[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                pf0[u] = make_double2(1.0, 0.0);
                pf1[u] = make_double2(1.0, 0.0);
                if (u < nb) {
                    if (W.has0) pf0[u] = logp[(r1 - 1 - u) * m + tid];
                    if (W.has1) pf1[u] = logp[(r1 - 1 - u) * m + P1];
                }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (u < nb) W.round(pf0[u], pf1[u]);
        }
        W.store(yc);                                    // back into the buffer it came from
    }
    __syncthreads();
    VI_STAMP(6);
    for (int s = tid; s < Np; s += JBS) {
        const int o = slot_orig0(s, m);
        if (o < N) C[sys * N + o] = yc[s];
    }
}


// Eigenvectors from the rotation log: V = J_1 J_2 ... J_K, so column k of V is the reverse replay applied to
// the unit vector of (final) slot k.  One workgroup per (system, block of CW columns); the CW vectors sit in
// LDS as G[slot][CW]; rotations are staged from HBM RB rounds at a time.  Output is the LAPACK/rocSOLVER
// layout Vout[k*N + r] (eigenvector k contiguous), k = final slot, r = original index.  Even N only.
template <int CW>
__global__ __launch_bounds__(JBS) void k_jacobi_vectors(int N, const double2* __restrict__ rotlog, int64_t log_stride,
                                                        const int* __restrict__ nround_in, double* __restrict__ Vout)
{
    constexpr int RB = 16;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int m = N >> 1;
    double* G = reinterpret_cast<double*>(lds_raw);                 // [N][CW]
    double2* st = reinterpret_cast<double2*>(G + (size_t)N * CW);    // [RB][m]
    const int tid = threadIdx.x;
    const int64_t sys = blockIdx.x;
    const int c0 = blockIdx.y * CW;
    const double2* logp = rotlog + sys * log_stride;
    const int nround = nround_in[sys];
    for (int i = tid; i < N * CW; i += JBS) {
        const int s = i / CW, c = i - s * CW;
        G[i] = (s == c0 + c) ? 1.0 : 0.0;
    }
    constexpr int ITV = 8;                                           // items per thread: m*CW <= ITV*JBS
    const int nitems = m * CW;
    int iP[ITV], ic[ITV];
#pragma unroll
    for (int it = 0; it < ITV; ++it) {
        const int i = tid + it * JBS;
        iP[it] = i < nitems ? i / CW : -1;
        ic[it] = i - (i / CW) * CW;
    }
    __syncthreads();
    for (int r1 = nround; r1 > 0; r1 -= RB) {
        const int nb = r1 >= RB ? RB : r1;
        for (int i = tid; i < nb * m; i += JBS) {
            const int u = i / m, P = i - u * m;
            st[u * m + P] = logp[(int64_t)(r1 - 1 - u) * m + P];
        }
        __syncthreads();
        for (int u = 0; u < nb; ++u) {
            double gp[ITV], gq[ITV];
#pragma unroll
            for (int it = 0; it < ITV; ++it)
                if (iP[it] >= 0) {
                    const int P = iP[it];
                    gp[it] = G[slot_next(2 * P, m) * CW + ic[it]];
                    gq[it] = G[slot_next(2 * P + 1, m) * CW + ic[it]];
                }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITV; ++it)
                if (iP[it] >= 0) {
                    const int P = iP[it];
                    const double2 r = st[u * m + P];
                    G[(2 * P) * CW + ic[it]] = r.x * gp[it] + r.y * gq[it];
                    G[(2 * P + 1) * CW + ic[it]] = -r.y * gp[it] + r.x * gq[it];
                }
            __syncthreads();
        }
    }
    double* Vo = Vout + sys * (int64_t)N * N;
    for (int i = tid; i < N * CW; i += JBS) {
        const int c = i / N, s = i - c * N;                          // s fastest: contiguous stores
        if (c0 + c < N) Vo[(int64_t)(c0 + c) * N + slot_orig0(s, m)] = G[s * CW + c];
    }
}

// Eigenvectors by register replay: one WAVE per strip of CPW columns, the columns in the wave's registers (WaveReplay)
// - no LDS, no barrier.  CPW = 1 for a single system (N independent waves spread over as many CUs: latency); CPW = 8 for
// batches, so that a system's rotation log is read N / 8 times instead of N times.  Even N <= 256 only.
template <int CPW>
__global__ __launch_bounds__(64) void k_jacobi_vectors_wave(int N, const double2* __restrict__ rotlog, int64_t log_stride,
                                                            const int* __restrict__ nround_in, double* __restrict__ Vout)
{
    const int m = N >> 1;
    const int lane = threadIdx.x;
    const int64_t sys = blockIdx.y;
    const int col0 = blockIdx.x * CPW;                  // final slots whose eigenvectors this wave builds
    const double2* logp = rotlog + sys * log_stride;
    const int64_t nround = nround_in[sys];
    WaveReplay W[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int col = col0 + c;
        W[c].init(lane, m);
        W[c].t0 = (2 * lane == col) ? 1.0 : 0.0;
        W[c].b0 = (2 * lane + 1 == col) ? 1.0 : 0.0;
        W[c].t1 = (2 * (lane + 64) == col) ? 1.0 : 0.0;
        W[c].b1 = (2 * (lane + 64) + 1 == col) ? 1.0 : 0.0;
    }
    constexpr int PF = 8;                               // rounds of (c, s) prefetched per batch
    const bool has0 = lane < m, has1 = lane + 64 < m;
    for (int64_t r1 = nround; r1 > 0; r1 -= PF) {
        const int nb = r1 >= PF ? PF : (int)r1;
        double2 pf0[PF], pf1[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            pf0[u] = make_double2(1.0, 0.0);
            pf1[u] = make_double2(1.0, 0.0);
            if (u < nb) {
                if (has0) pf0[u] = logp[(r1 - 1 - u) * m + lane];
                if (has1) pf1[u] = logp[(r1 - 1 - u) * m + lane + 64];
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (u < nb) {
#pragma unroll
                for (int c = 0; c < CPW; ++c) W[c].round(pf0[u], pf1[u]);
            }
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int col = col0 + c;
        if (col >= N) break;
        double* Vo = Vout + sys * (int64_t)N * N + (int64_t)col * N;     // eigenvector `col`, indexed by original index
        if (has0) {
            Vo[slot_orig0(2 * lane, m)] = W[c].t0;
            Vo[slot_orig0(2 * lane + 1, m)] = W[c].b0;
        }
        if (has1) {
            Vo[slot_orig0(2 * (lane + 64), m)] = W[c].t1;
            Vo[slot_orig0(2 * (lane + 64) + 1, m)] = W[c].b1;
        }
    }
}

}  // namespace

#ifdef VI_STAMPS
extern "C" int vi_debug_jacobi_rot(int* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_jacobi_rot), 64 * sizeof(int)) == hipSuccess ? 0 : -1;
}

extern "C" int vi_debug_jacobi_stamps(double* out, int reset)
{
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_jacobi_stamps), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) out[i] = (double)h[i];
    if (reset) {
        memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_jacobi_stamps), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

size_t vi_jacobi_lds_bytes(int N)
{
    const int Np = (N + 1) & ~1, m = Np / 2;
    const int nt = Np * (Np + 1) / 2;
    const int nblk = m * (m - 1) / 2;
    int iters = (nblk + JBS - 1) / JBS;
    iters = iters <= 1 ? 1 : iters <= 2 ? 2 : iters <= 5 ? 5 : 10;      // the instantiated IT (vi_jacobi_solve)
    size_t b = (size_t)(nt + 2 * Np + (nt & 1)) * 8 + (size_t)m * 16 + 8 * 8;
    b += (size_t)4 * (iters * JBS - nblk) * 8;       // scratch quadruples of the idle block slots
    return (b + 15) & ~(size_t)15;
}

static int jacobi_iters(int N)
{
    const int m = ((N + 1) & ~1) / 2;
    return (m * (m - 1) / 2 + JBS - 1) / JBS;
}

bool vi_jacobi_supported(int N)
{
    return N >= 8 && vi_jacobi_lds_bytes(N) <= 159 * 1024 && ((N + 1) / 2) <= JBS && jacobi_iters(N) <= 10;
}

size_t vi_jacobi_log_bytes(int N, int max_sweeps)
{
    const int Np = (N + 1) & ~1, m = Np / 2;
    return (size_t)max_sweeps * (Np - 1) * m * sizeof(double2);
}

template <int IT>
static int launch_jacobi(vi_ctx* c, int64_t B, int N, const double* d_X, const double* d_scl, const double* d_y,
                         const int* d_rec, double rcond, double* d_C, int* d_rank, void* d_log, int max_sweeps,
                         int* d_sweeps, double* d_lam, int lam_raw, int* d_nround, double abs_floor)
{
    const size_t shm = vi_jacobi_lds_bytes(N);
    static size_t attr_max = 0;
    if (shm > attr_max) {
        VI_HIP(hipFuncSetAttribute((const void*)k_jacobi_solve<IT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        attr_max = shm;
    }
    const int Np = (N + 1) & ~1, m = Np / 2;
    const int64_t log_stride = (int64_t)max_sweeps * (Np - 1) * m;
    // abs_floor: the systems are scaled to max|X| in [1, 2).  Cold solves pass 1e-22: off-diagonal elements
    // below it cannot move any kept eigenvalue (|lambda| > eps * max|lambda|) by more than 1e-6 of itself.
    // Warm solves pass 1e-16: the rotated system D1 + alpha D2 carries formation errors of N*eps anyway.
    const int slot = (int)(c->solve_launches % vi_ctx::NSOLVE_EV);
    if (c->solve_timing) VI_HIP(hipEventRecord(c->evs[slot][0], c->stream));
    hipLaunchKernelGGL(k_jacobi_solve<IT>, dim3((unsigned)B), dim3(JBS), shm, c->stream, N, d_X, d_scl, d_y, d_rec,
                       rcond, abs_floor, d_C, d_rank, (double2*)d_log, log_stride, max_sweeps, d_sweeps, d_lam, lam_raw, d_nround,
                       c->solve_timing ? c->d_rounds : nullptr);
    VI_HIP(hipGetLastError());
    if (c->solve_timing) {
        VI_HIP(hipEventRecord(c->evs[slot][1], c->stream));
        c->solve_launches += 1;
        c->solve_systems += B;
    }
    return VI_OK;
}

// d_X: systems scaled by k_scale_system (B x N x N, only read).
int vi_jacobi_solve(vi_ctx* c, int64_t B, int N, const double* d_X, const double* d_scl, const double* d_y,
                    const int* d_rec, double rcond, double* d_C, int* d_rank, void* d_log, int max_sweeps,
                    int* d_sweeps, double* d_lam, int lam_raw, int* d_nround, double abs_floor)
{
    const int it = jacobi_iters(N);
#define VI_J(IT) return launch_jacobi<IT>(c, B, N, d_X, d_scl, d_y, d_rec, rcond, d_C, d_rank, d_log, max_sweeps, d_sweeps, d_lam, lam_raw, d_nround, abs_floor)
    if (it <= 1) VI_J(1);
    if (it <= 2) VI_J(2);
    if (it <= 5) VI_J(5);
    VI_J(10);
#undef VI_J
}

// Eigenvectors (column k = eigenvector of slot k, LAPACK layout) from the rotation logs of vi_jacobi_solve.
bool vi_jacobi_vectors_supported(int N) { return vi_jacobi_supported(N) && (N % 2 == 0) && N <= 192; }

template <int CW>
static int launch_vectors(vi_ctx* c, int64_t B, int N, const void* d_log, int max_sweeps, const int* d_nround, double* d_V)
{
    const int m = N / 2;
    if (m * CW > 8 * JBS) {
        vi_set_error("vi_jacobi_vectors: N=%d too large", N);
        return VI_ERR_UNSUPPORTED;
    }
    const size_t shm = (size_t)N * CW * sizeof(double) + (size_t)16 * m * sizeof(double2);
    static size_t attr_max = 0;
    if (shm > attr_max) {
        VI_HIP(hipFuncSetAttribute((const void*)k_jacobi_vectors<CW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        attr_max = shm;
    }
    const int64_t log_stride = (int64_t)max_sweeps * (N - 1) * m;
    hipLaunchKernelGGL(k_jacobi_vectors<CW>, dim3((unsigned)B, (unsigned)((N + CW - 1) / CW)), dim3(JBS), shm, c->stream, N,
                       (const double2*)d_log, log_stride, d_nround, d_V);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

int vi_jacobi_vectors(vi_ctx* c, int64_t B, int N, const void* d_log, int max_sweeps, const int* d_nround, double* d_V)
{
    // very few systems (one record's prepare / final solve): one wave per column with the column in registers
    // (measured at N = 144, one system: 1.47 ms for the workgroup kernel below, 0.62 ms for one column per wave through
    // LDS, and the register version after that);
    // few systems: narrow column blocks spread one system over more CUs (latency); many systems: wide blocks
    // re-read the rotation log fewer times (throughput)
    static int wave_ok = -1;
    if (wave_ok < 0) {
        const char* e = getenv("VINTERP_VECTORS");
        wave_ok = (e && !strcmp(e, "block")) ? 0 : 1;
    }
    if (wave_ok && N <= 256 && (N % 2) == 0) {
        const int m = N / 2;
        const int64_t log_stride = (int64_t)max_sweeps * (N - 1) * m;
        // the strips of one system are neighbours in the grid (x fastest), so they read its log together through L2
        if (B * N <= 2 * c->n_cu)
            hipLaunchKernelGGL(k_jacobi_vectors_wave<1>, dim3((unsigned)N, (unsigned)B), dim3(64), 0, c->stream, N,
                               (const double2*)d_log, log_stride, d_nround, d_V);
        else
            hipLaunchKernelGGL(k_jacobi_vectors_wave<8>, dim3((unsigned)((N + 7) / 8), (unsigned)B), dim3(64), 0, c->stream, N,
                               (const double2*)d_log, log_stride, d_nround, d_V);
        VI_HIP(hipGetLastError());
        return VI_OK;
    }
    if (B * ((N + 35) / 36) < c->n_cu / 2) return launch_vectors<12>(c, B, N, d_log, max_sweeps, d_nround, d_V);
    return launch_vectors<36>(c, B, N, d_log, max_sweeps, d_nround, d_V);
}
