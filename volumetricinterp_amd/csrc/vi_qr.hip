// K3p: orthogonal PRE-CONDITIONING of the truncated solve by one column-pivoted Householder QR step.
//
// The systems X(alpha) = A^T W A + alpha R of the regularisation-parameter search (volumetricinterp/interpolate.py:460-462)
// are graded over ~50 decades and rank deficient; a cyclic Jacobi iteration started from X itself spends 12-16 of its
// 20-24 sweeps in a linear phase in which every pair rotates and the grading is sorted out one scale at a time.  With
// X P = Q R (Householder QR, columns pivoted by remaining norm) the similar matrix X1 = Q^T X Q = R P^T Q has its scales
// concentrated row by row, and the same Jacobi kernel converges on it in 5-8 sweeps - and, the rotations being far fewer
// and better scaled, reproduces 50-digit arithmetic to 1e-10 in chi^2 and 1e-7 in the coefficients where the plain
// iteration reaches 1e-5 / 4e-3 (tools/exp_qr_precond.py, tests/golden/exact_default_c2.npz).  The solve becomes
//     X1 = Q^T X Q,  y1 = Q^T y   (k_qr_sim, this file)  ->  X1 c1 = y1  (k_jacobi_solve)  ->  c = Q c1  (k_qr_back_*).
// (Static column orders - by initial norm or by diagonal - give 9 sweeps instead of 7, unpivoted QR none of the gain.)
//
// k_qr_sim: one workgroup per system, the matrix in REGISTERS - a 144 x 144 fp64 matrix (166 KB) does not fit the 160 KB
// of LDS.  An "octet" of eight lanes holds two columns, lane l their rows 8 i + l (18 local rows at N = 144: 36 doubles
// per thread); y rides along as column N.  LDS holds the current Householder vector and the packed store of all of them.
// A reflector costs every thread two LDS reads per local row (dot product, update) for two columns - with four lanes
// per column it was twice the LDS traffic, and the LDS pipe (128 B / clk for the whole CU), not the arithmetic, set the
// pace: 4400 cycles per update (tools/exp_qr_stamps.py).
//   phase 1 (N - 1 steps, two LDS-only barriers each): pivot = the live column of largest remaining norm (norms are
//     recomputed in the update pass of the previous step, not down-dated: the matrices are graded); its owner forms
//     v = x - alpha e_k, tau = 2 / v^T v; every column takes c -= tau (v^T c) v.  Row k of the updated matrix - row k
//     of R P^T - then leaves the registers for global memory.
//   phase 2 (no barriers): X1 = Q^T X Q = (R P^T) Q, and X1 is symmetric, so its column j is Q^T applied to row j of
//     R P^T: every octet loads its two rows and takes all reflectors; y1 = Q^T y the same way.
// The reflectors leave in packed form (v_k, then the N taus) for the back-transformation kernels.
#include "vi_common.h"

#include <type_traits>

#ifdef VI_STAMPS
// diagnostic build only (-DVI_STAMPS): cycle sums of thread 0 of workgroup 0 per part of k_qr_sim; vi_debug_qr_stamps
__device__ unsigned long long g_qr_stamps[8];
#define QR_STAMP(k)                                                        \
    do {                                                                   \
        const unsigned long long t_ = __builtin_readcyclecounter();        \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_qr_stamps[k] += t_ - stamp_t; \
        stamp_t = __builtin_readcyclecounter();                            \
    } while (0)
#else
#define QR_STAMP(k)
#endif

namespace {

template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov(double v)      // lanes without a source keep their own value
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the eight lanes of an octet, the same bits in every lane: quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_half_mirror (lane l <-> 7 - l)
__device__ __forceinline__ double octet_sum(double v)
{
    v += dpp_mov<0xB1, 0xf>(v);
    v += dpp_mov<0x4E, 0xf>(v);
    v += dpp_mov<0x141, 0xf>(v);
    return v;
}
// max over the wave of a value that is uniform within each octet, valid in every lane: row_shr:8 inside the rows of 16
// lanes, row_bcast:15 / row_bcast:31 across them, a read of lane 63 - 6 DPP moves instead of the dependent ds_bpermute
// round trips of a shuffle butterfly (the reduction sits on the critical path of every step).
__device__ __forceinline__ double wave_max_octets(double v)
{
    v = fmax(v, dpp_mov<0x118, 0xf>(v));
    v = fmax(v, dpp_mov<0x142, 0xa>(v));
    v = fmax(v, dpp_mov<0x143, 0xc>(v));
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// sum over the whole wave, valid in every lane
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_mov<0xB1, 0xf>(v);
    v += dpp_mov<0x4E, 0xf>(v);
    v += dpp_mov<0x141, 0xf>(v);
    v += dpp_mov<0x140, 0xf>(v);                                  // row_mirror: every lane now holds its row's sum
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x142, 0xa, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x142, 0xa, 0xf, false);
        v += __hiloint2double(hi, lo);                            // rows 1, 3 += rows 0, 2
    }
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x143, 0xc, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x143, 0xc, 0xf, false);
        v += __hiloint2double(hi, lo);                            // rows 2, 3 += rows 0 + 1
    }
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the global stores in flight (rows of
// R P^T leave for global memory in every step of the factorisation; nobody reads them before the phase is over), a
// round trip of ~1 us, twice per step.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 1 / sqrt(x), 1 / x for x well inside the normal range: hardware estimate + one third-order / two Newton steps
// (see fast_rsqrt in vi_jacobi.hip)
__device__ __forceinline__ double qr_rsqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y, e * fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ double qr_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    y = fma(y, fma(-x, y, 1.0), y);
    return y;
}

// Packed store of the reflectors.  v_k lives in rows k .. N-1; it is stored from row 8 (k >> 3) on (zeros above k) up to
// row NR - 1 (NR = 8 x the local rows of the kernel instantiation that serves N; zeros below N), so that an octet takes
// it in whole local rows without predicates.
__host__ __device__ __forceinline__ int hh_off(int k, int NR)
{
    const int B = k >> 3;
    return 8 * NR * B - 32 * B * (B - 1) + (k & 7) * (NR - 8 * B);
}
__host__ __device__ __forceinline__ size_t hh_doubles(int N, int NR)
{
    return (size_t)hh_off(N, NR) + (size_t)N;               // N vectors (the last one unused), then the N taus
}

// compile-time loops over the local row the current reflector starts in: the body is instantiated once per value, so
// every register index is static and a step is straight-line code (with run-time row tests the unrolled loops became a
// web of small blocks whose register copies and spills cost more than the arithmetic)
template <int I, int E, class F>
__device__ __forceinline__ bool static_up(F&& f)          // f(I), f(I+1), ... until one returns true
{
    if constexpr (I < E) {
        if (f(std::integral_constant<int, I>{})) return true;
        return static_up<I + 1, E>(f);
    }
    return false;
}
template <int I, class F>
__device__ __forceinline__ void static_down(F&& f)        // f(I), f(I-1), ..., f(0)
{
    if constexpr (I >= 0) {
        f(std::integral_constant<int, I>{});
        static_down<I - 1>(f);
    }
}

// one reflector on the two columns of an octet: c -= tau (v^T c) v, vl = &v[l] with v valid (zero-padded) from row
// 8 I0 (I0 = k >> 3, static) to row 8 RT - 1 of the LDS image.
template <int RT, int I0>
__device__ __forceinline__ void reflect2(double (&a0)[RT], double (&a1)[RT], const double* __restrict__ vl, double tau)
{
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int i = I0; i < RT; ++i) {
        const double v = vl[8 * i];
        d0 = fma(v, a0[i], d0);
        d1 = fma(v, a1[i], d1);
    }
    const double w0 = tau * octet_sum(d0), w1 = tau * octet_sum(d1);
    // the vector is read again rather than kept: 36 more registers do not fit beside two columns (168 at three waves per SIMD)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = I0; i < RT; ++i) {
        const double v = vl[8 * i];
        a0[i] = fma(-w0, v, a0[i]);
        a1[i] = fma(-w1, v, a1[i]);
    }
}

template <int RT>
__global__ __launch_bounds__(RT <= 8 ? 320 : 640) void k_qr_sim(
    int N, const double* __restrict__ X, const double* __restrict__ y, const int* __restrict__ rec, double* X1,
    double* __restrict__ y1, double* __restrict__ hh, int64_t hh_stride, double* __restrict__ Rscr)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int NR = 8 * RT;                                           // rows an octet can hold
    double* vcur = reinterpret_cast<double*>(lds_raw);                   // [2][NR] current reflector (double-buffered)
    double* Vst = vcur + 2 * NR;                                         // packed reflectors
    const int nvst = hh_off(N, NR);
    double* taus = Vst + nvst;                                           // [N]
    double* redv = taus + N;                                             // [16]
    int* redi = reinterpret_cast<int*>(redv + 16);                       // [16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int l = tid & 7, c0 = (tid >> 3) << 1, c1 = c0 + 1;
    const int64_t sys = blockIdx.x;
    const double* Xs = X + sys * (int64_t)N * N;
    const double* ys = y + (int64_t)(rec ? rec[sys] : sys) * N;
    double* Rs = Rscr + sys * (int64_t)N * N;
    const bool m0 = c0 < N, m1 = c1 < N;                                 // matrix columns (column N is y)

#ifdef VI_STAMPS
    unsigned long long stamp_t = __builtin_readcyclecounter();
#endif
    double a0[RT], a1[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int r = 8 * i + l;
        // the lower triangle mirrored: exactly symmetric, and the elements the plain Jacobi kernel reads
        a0[i] = (r < N) ? (m0 ? Xs[(int64_t)(r > c0 ? r : c0) * N + (r > c0 ? c0 : r)] : (c0 == N ? ys[r] : 0.0)) : 0.0;
        a1[i] = (r < N) ? (m1 ? Xs[(int64_t)(r > c1 ? r : c1) * N + (r > c1 ? c1 : r)] : (c1 == N ? ys[r] : 0.0)) : 0.0;
    }
    for (int r = tid; r < 2 * NR; r += blockDim.x) vcur[r] = 0.0;
    for (int k = tid; k < N; k += blockDim.x) taus[k] = 0.0;
    for (int e = tid; e < nvst; e += blockDim.x) Vst[e] = 0.0;
    double n0 = 0.0, n1 = 0.0;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        n0 = fma(a0[i], a0[i], n0);
        n1 = fma(a1[i], a1[i], n1);
    }
    n0 = octet_sum(n0);
    n1 = octet_sum(n1);
    bool done0 = !m0, done1 = !m1;
    int K = 0;
    __syncthreads();
    QR_STAMP(0);
    // ---- phase 1: pivoted Householder QR.  Invariant at the top of step k: rows < k of every column are zero in the
    //      registers (row k' went to Rs[k'][.] = (R P^T)[k'][.] when step k' finished), n0 / n1 = remaining norms^2.
    static_up<0, RT>([&](auto I0c) -> bool {
        constexpr int I0 = decltype(I0c)::value;             // local row of the steps k = 8 I0 .. 8 I0 + 7
        for (int lk = 0; lk < 8; ++lk) {                      // row k = local row I0 of lane lk
            const int k = 8 * I0 + lk;
            if (k >= N - 1) return true;
            const double v0 = done0 ? -1.0 : n0, v1 = done1 ? -1.0 : n1;
            const double cv = v1 > v0 ? v1 : v0;                      // ties: the lower column
            const int ci = v1 > v0 ? c1 : c0;
            const double wm = wave_max_octets(cv);
            const unsigned long long hit = __ballot(cv >= 0.0 && cv == wm);
            const int first = hit ? (__ffsll((long long)hit) - 1) : 0;
            const int wi = __builtin_amdgcn_readlane(ci, first);
            if (lane == 0) {
                redv[wave] = hit ? wm : -1.0;
                redi[wave] = wi;
            }
            lds_barrier();
            QR_STAMP(1);
            double pmax = redv[0];
            int p = redi[0];
#pragma unroll
            for (int w = 1; w < 10; ++w) {
                const double ov = w < nw ? redv[w] : -1.0;
                const int oi = w < nw ? redi[w] : 0;
                const bool take = ov > pmax;                 // ties: the first wave, i.e. the smallest column index
                pmax = take ? ov : pmax;
                p = take ? oi : p;
            }
            if (!(pmax > 0.0)) return true;          // what is left is exactly zero: the remaining reflectors are identities
            double* vb = vcur + (k & 1) * NR + l;
            if ((p >> 1) == (tid >> 3)) {
                const bool second = p & 1;
                const double xk = __shfl(second ? a1[I0] : a0[I0], (lane & ~7) | lk);
                const double ri_ = qr_rsqrt(pmax);
                const double nx = pmax * ri_;                               // ||x||
                const double alpha = -copysign(nx, xk);                     // H x = alpha e_k
                const double vk = xk - alpha;                               // no cancellation: |vk| = |xk| + ||x||
                const double tau = ri_ * qr_rcp(nx + fabs(xk));             // 2 / v^T v = 1 / (||x|| (||x|| + |xk|))
                double* vs = Vst + hh_off(k, NR) - 8 * I0 + l;
#pragma unroll
                for (int i = I0; i < RT; ++i) {
                    const double x = second ? a1[i] : a0[i];                // rows < k are zero
                    const double v = (i == I0 && l == lk) ? vk : x;
                    vb[8 * i] = v;
                    vs[8 * i] = v;
                }
                if (l == 0) taus[k] = tau;
                done0 = done0 || !second;
                done1 = done1 || second;
            }
            lds_barrier();
            QR_STAMP(2);
            K = k + 1;
            const double tau = taus[k];
            double d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int i = I0; i < RT; ++i) {
                const double v = vb[8 * i];
                d0 = fma(v, a0[i], d0);
                d1 = fma(v, a1[i], d1);
            }
            const double w0 = tau * octet_sum(d0), w1 = tau * octet_sum(d1);
            asm volatile("" ::: "memory");                // read v again (see reflect2)
            n0 = 0.0;
            n1 = 0.0;
#pragma unroll
            for (int i = I0; i < RT; ++i) {
                const double v = vb[8 * i];
                a0[i] = fma(-w0, v, a0[i]);
                a1[i] = fma(-w1, v, a1[i]);
                if (i == I0) {                    // row k is finished: it leaves for (R P^T)[k][.]
                    if (l == lk) {
                        if (m0) Rs[(int64_t)k * N + c0] = a0[i];
                        if (m1) Rs[(int64_t)k * N + c1] = a1[i];
                        a0[i] = 0.0;
                        a1[i] = 0.0;
                    }
                }
                n0 = fma(a0[i], a0[i], n0);
                n1 = fma(a1[i], a1[i], n1);
            }
            n0 = octet_sum(n0);
            n1 = octet_sum(n1);
            QR_STAMP(3);
        }
        return false;
    });
    __syncthreads();
    // rows K .. N-1 of R P^T (what is left in the registers when the loop ends: the last row, or zeros after a break)
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int r = 8 * i + l;
        if (r >= K && r < N) {
            if (m0) Rs[(int64_t)r * N + c0] = a0[i];
            if (m1) Rs[(int64_t)r * N + c1] = a1[i];
        }
    }
    {
        double* hs = hh + sys * hh_stride;
        for (int e = tid; e < nvst + N; e += blockDim.x) hs[e] = Vst[e];         // the taus follow the vectors in LDS as well
    }
    __syncthreads();
    QR_STAMP(4);
    // ---- phase 2: X1 = Q^T X Q = (R P^T) Q = (Q^T (R P^T)^T)^T, and X1 is symmetric: column c of (R P^T)^T is row c of
    //      Rs, contiguous; it takes all reflectors.  y1 = Q^T y the same way from y itself.
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int r = 8 * i + l;
        a0[i] = (r < N) ? (m0 ? Rs[(int64_t)c0 * N + r] : (c0 == N ? ys[r] : 0.0)) : 0.0;
        a1[i] = (r < N) ? (m1 ? Rs[(int64_t)c1 * N + r] : (c1 == N ? ys[r] : 0.0)) : 0.0;
    }
    QR_STAMP(5);
    static_up<0, RT>([&](auto I0c) -> bool {
        constexpr int I0 = decltype(I0c)::value;
        for (int k = 8 * I0; k < 8 * I0 + 8; ++k) {
            if (k >= K) return true;
            reflect2<RT, I0>(a0, a1, Vst + hh_off(k, NR) - 8 * I0 + l, taus[k]);
        }
        return false;
    });
    QR_STAMP(6);
    double* Xo = X1 + sys * (int64_t)N * N;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int r = 8 * i + l;
        if (r < N) {
            if (m0) Xo[(int64_t)c0 * N + r] = a0[i];
            else if (c0 == N) y1[sys * N + r] = a0[i];
            if (m1) Xo[(int64_t)c1 * N + r] = a1[i];
            else if (c1 == N) y1[sys * N + r] = a1[i];
        }
    }
}

// c <- Q c = H_0 H_1 ... H_{K-1} c for one vector per system: the reflectors are staged in LDS by the whole workgroup,
// then wave 0 applies them with the vector in its registers (element lane + 64 j).
template <int EPL>
__global__ __launch_bounds__(256) void k_qr_back_vec(int N, int NR, const double* __restrict__ hh, int64_t hh_stride,
                                                     double* __restrict__ C)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* Vst = reinterpret_cast<double*>(lds_raw);
    const int nvst = hh_off(N, NR);
    const double* taus = Vst + nvst;
    const int tid = threadIdx.x;
    const int64_t sys = blockIdx.x;
    const double* hs = hh + sys * hh_stride;
    for (int e = tid; e < nvst + N; e += blockDim.x) Vst[e] = hs[e];
    __syncthreads();
    if (tid >= 64) return;
    double* c = C + sys * N;
    double x[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int r = tid + 64 * j;
        x[j] = r < N ? c[r] : 0.0;
    }
    for (int k = N - 2; k >= 0; --k) {
        const double tau = taus[k];
        if (tau == 0.0) continue;                    // no reflector was formed at this step (the rest of the matrix was zero)
        const double* vk = Vst + hh_off(k, NR) - 8 * (k >> 3);
        const int r0 = 8 * (k >> 3);
        double v[EPL];
        double dot = 0.0;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int r = tid + 64 * j;
            v[j] = (r >= r0 && r < NR) ? vk[r] : 0.0;
            dot = fma(v[j], x[j], dot);
        }
        const double w = tau * wave_sum(dot);
#pragma unroll
        for (int j = 0; j < EPL; ++j) x[j] = fma(-w, v[j], x[j]);
    }
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int r = tid + 64 * j;
        if (r < N) c[r] = x[j];
    }
}

// V[:, j] <- Q V[:, j] for the N vectors of a system (eigenvectors in the LAPACK layout: vector j contiguous): one
// workgroup per system, reflectors staged in LDS, two vectors per octet.
template <int RT>
__global__ __launch_bounds__(RT <= 8 ? 320 : 640) void k_qr_back_mat(int N, const double* __restrict__ hh, int64_t hh_stride,
                                                                     double* __restrict__ V)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* Vst = reinterpret_cast<double*>(lds_raw);
    constexpr int NR = 8 * RT;
    const int nvst = hh_off(N, NR);
    double* taus = Vst + nvst;
    const int tid = threadIdx.x, l = tid & 7, c0 = (tid >> 3) << 1, c1 = c0 + 1;
    const int64_t sys = blockIdx.x;
    const double* hs = hh + sys * hh_stride;
    for (int e = tid; e < nvst + N; e += blockDim.x) Vst[e] = hs[e];
    __syncthreads();
    if (c0 >= N) return;
    double* Vs = V + sys * (int64_t)N * N;
    const bool m1 = c1 < N;
    double a0[RT], a1[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int r = 8 * i + l;
        a0[i] = r < N ? Vs[(int64_t)c0 * N + r] : 0.0;
        a1[i] = (r < N && m1) ? Vs[(int64_t)c1 * N + r] : 0.0;
    }
    static_down<RT - 1>([&](auto I0c) {
        constexpr int I0 = decltype(I0c)::value;
        for (int k = 8 * I0 + 7; k >= 8 * I0; --k) {
            const double tau = k < N ? taus[k] : 0.0;
            if (tau != 0.0) reflect2<RT, I0>(a0, a1, Vst + hh_off(k, NR) - 8 * I0 + l, tau);
        }
    });
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int r = 8 * i + l;
        if (r < N) {
            Vs[(int64_t)c0 * N + r] = a0[i];
            if (m1) Vs[(int64_t)c1 * N + r] = a1[i];
        }
    }
}

}  // namespace
#ifdef VI_STAMPS
extern "C" int vi_debug_qr_stamps(double* out, int reset)
{
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_qr_stamps), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) out[i] = (double)h[i];
    if (reset) {
        memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_qr_stamps), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
namespace {

// local rows per lane (eight lanes per pair of columns): the smallest instantiation that holds N rows
int qr_rt(int N)
{
    const int need = (N + 7) / 8;
    const int opts[] = {4, 6, 8, 12, 18};
    for (int o : opts)
        if (need <= o) return o;
    return 0;
}

size_t qr_lds_bytes(int N)
{
    return (size_t)2 * 8 * qr_rt(N) * 8 + (hh_doubles(N, 8 * qr_rt(N)) + 16) * 8 + 16 * 4;
}

}  // namespace

bool vi_qr_supported(int N) { return N >= 8 && qr_rt(N) > 0 && qr_lds_bytes(N) <= 159 * 1024; }
size_t vi_qr_hh_bytes(int N) { return hh_doubles(N, 8 * qr_rt(N)) * sizeof(double); }

// d_X: B scaled symmetric systems (read once, not written unless d_X1 == d_X); d_y (+ d_rec) as in vi_jacobi_solve.
// Outputs: d_X1 (B x N x N; may alias d_X), d_y1 (B x N), d_hh (B x vi_qr_hh_bytes(N)); d_scr: B x N x N scratch.
// hh_stride: distance in doubles between the reflector sets of consecutive systems (0: back to back).
int vi_qr_precond(vi_ctx* c, int64_t B, int N, const double* d_X, const double* d_y, const int* d_rec, double* d_X1,
                  double* d_y1, double* d_hh, double* d_scr, int64_t hh_stride)
{
    if (hh_stride <= 0) hh_stride = (int64_t)(vi_qr_hh_bytes(N) / sizeof(double));
    const int rt = qr_rt(N);
    const int threads = ((8 * ((N + 2) / 2) + 63) / 64) * 64;           // an octet per pair of columns, y is column N
    const size_t shm = qr_lds_bytes(N);
#define VI_QR(R)                                                                                                            \
    do {                                                                                                                    \
        VI_HIP(hipFuncSetAttribute((const void*)k_qr_sim<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));        \
        hipLaunchKernelGGL(k_qr_sim<R>, dim3((unsigned)B), dim3(threads), shm, c->stream, N, d_X, d_y, d_rec, d_X1, d_y1,   \
                           d_hh, hh_stride, d_scr);                                                                         \
    } while (0)
    switch (rt) {
    case 4: VI_QR(4); break;
    case 6: VI_QR(6); break;
    case 8: VI_QR(8); break;
    case 12: VI_QR(12); break;
    case 18: VI_QR(18); break;
    default: vi_set_error("vi_qr_precond: N=%d unsupported", N); return VI_ERR_UNSUPPORTED;
    }
#undef VI_QR
    VI_HIP(hipGetLastError());
    return VI_OK;
}

int vi_qr_back_vec(vi_ctx* c, int64_t B, int N, const double* d_hh, double* d_C, int64_t hh_stride)
{
    if (hh_stride <= 0) hh_stride = (int64_t)(vi_qr_hh_bytes(N) / sizeof(double));
    const size_t shm = vi_qr_hh_bytes(N);
#define VI_QV(E)                                                                                                            \
    do {                                                                                                                    \
        VI_HIP(hipFuncSetAttribute((const void*)k_qr_back_vec<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));   \
        hipLaunchKernelGGL(k_qr_back_vec<E>, dim3((unsigned)B), dim3(256), shm, c->stream, N, 8 * qr_rt(N), d_hh, hh_stride, d_C);        \
    } while (0)
    if (N <= 64) VI_QV(1);
    else if (N <= 128) VI_QV(2);
    else VI_QV(3);
#undef VI_QV
    VI_HIP(hipGetLastError());
    return VI_OK;
}

int vi_qr_back_mat(vi_ctx* c, int64_t B, int N, const double* d_hh, double* d_V, int64_t hh_stride)
{
    if (hh_stride <= 0) hh_stride = (int64_t)(vi_qr_hh_bytes(N) / sizeof(double));
    const int rt = qr_rt(N);
    const int threads = ((8 * ((N + 1) / 2) + 63) / 64) * 64;
    const size_t shm = vi_qr_hh_bytes(N);
#define VI_QB(R)                                                                                                            \
    do {                                                                                                                    \
        VI_HIP(hipFuncSetAttribute((const void*)k_qr_back_mat<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));   \
        hipLaunchKernelGGL(k_qr_back_mat<R>, dim3((unsigned)B), dim3(threads), shm, c->stream, N, d_hh, hh_stride, d_V);    \
    } while (0)
    switch (rt) {
    case 4: VI_QB(4); break;
    case 6: VI_QB(6); break;
    case 8: VI_QB(8); break;
    case 12: VI_QB(12); break;
    case 18: VI_QB(18); break;
    default: vi_set_error("vi_qr_back_mat: N=%d unsupported", N); return VI_ERR_UNSUPPORTED;
    }
#undef VI_QB
    VI_HIP(hipGetLastError());
    return VI_OK;
}

// Stage-test entry (tests/test_gpu_search_stages.py): X1 = Q^T X Q and the explicit Q of B symmetric systems.
// d_X is only read; d_X1, d_Q: B x N x N (Q in the LAPACK layout of eigenvector matrices: Q[:, j] contiguous).
extern "C" int vi_qr_similarity_f64(vi_ctx* c, int64_t B, int32_t N, const double* d_X, const double* d_y, double* d_X1,
                                    double* d_y1, double* d_Q)
{
    VI_REQUIRE(c && d_X && d_y && d_X1 && d_y1 && d_Q, "null argument");
    VI_REQUIRE(B >= 0 && N > 0, "bad size");
    if (B == 0) return VI_OK;
    if (!vi_qr_supported(N)) {
        vi_set_error("vi_qr_similarity_f64: N=%d outside the range of the register-resident QR", N);
        return VI_ERR_UNSUPPORTED;
    }
    VI_HIP(hipSetDevice(c->device));
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, (size_t)B * (vi_qr_hh_bytes(N) + (size_t)N * N * sizeof(double)) + 256, &ws);
    if (rc != VI_OK) return rc;
    double* hhp = (double*)ws;
    double* scr = hhp + (size_t)B * (vi_qr_hh_bytes(N) / sizeof(double));
    if ((rc = vi_qr_precond(c, B, N, d_X, d_y, nullptr, d_X1, d_y1, hhp, scr, 0)) != VI_OK) return rc;
    // explicit Q: the back-transformation of the identity
    std::vector<double> eye((size_t)N * N, 0.0);
    for (int i = 0; i < N; ++i) eye[(size_t)i * N + i] = 1.0;
    for (int64_t b = 0; b < B; ++b)
        VI_HIP(hipMemcpyAsync(d_Q + b * (int64_t)N * N, eye.data(), eye.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    VI_HIP(hipStreamSynchronize(c->stream));
    return vi_qr_back_mat(c, B, N, hhp, d_Q, 0);
}
